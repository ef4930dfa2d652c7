// Decode-step projection GEMMs (M = batch rows, one new token per image):  out = f(A) * W^T with the
// LayerNorm sandwich fused in front and the block's non-linearity / residual fused behind.
//
// Reference per decode position (KV-cached form of model/decoder.py:41-67 + model/attention.py:242-259):
//   PRO_EMBED : x = tok_emb[token] + pos_emb[t]           (decoder.py:51-52)  ; z = LN(x)
//   PRO_LN2   : x = LN(y) (next residual) ; z = LN(x)     (attention.py:257-259 then :243, ONE shared LN)
//   PRO_LNF   : z = LN_final(y)                           (decoder.py:57)
//   PRO_NONE  : A read as is (attention output / FFN hidden)
//   EPI_QKV   : q -> qbuf, k/v appended to the self-attention cache at position t   (attention.py:124-127)
//   EPI_Q     : q -> qbuf                                  (cross attention; K/V were projected once)
//   EPI_GLU_RES : y = (v+b)*sigmoid(g+b) + residual        (attention.py:96-99,180 ; Residual :35-38)
//   EPI_GEGLU : h = (v+b)*gelu_erf(g+b)                    (attention.py:15-17)
//   EPI_BIAS_RES: y = acc + b + residual                   (attention.py:66, :35-38)
//   EPI_LOGITS: logits = acc + b                           (decoder.py:60, last position only)
//   EPI_STORE_T: out = acc in the storage type -> h_out [rows][F]   (latent attention: the q / q' projection, lat_attn.h; with z_cache the
//                normalised input rows are appended to the self attention's latent history)
//
// These launches are latency chains, not throughput kernels (a few MFLOP each), so the structure minimises
// dependent memory round trips: output tile = 16 rows x 32 columns per 256-thread block (many small blocks
// -> every block's weight slice is 32 rows x K, read once, straight from L2/MALL into VGPRs); the four
// waves split K in interleaved 64-byte chunks and each wave issues ALL of its weight/activation fragment
// loads before the first MFMA; the LayerNorm prologue handles its 16 rows in one pass (16 lanes per row,
// gamma/beta/bias/residual prefetched at kernel entry); partial tiles are summed through LDS and the
// epilogue is spread over all 256 threads.  The decode position t is read from device memory so that one
// captured launch sequence can be replayed for every step.
// Bound: latency / L2 weight streaming.
#pragma once
#include "common.h"

namespace txo {

enum { PRO_NONE = 0, PRO_EMBED = 1, PRO_LN2 = 2, PRO_LNF = 3 };
enum { EPI_QKV = 0, EPI_Q = 1, EPI_GLU_RES = 2, EPI_GEGLU = 3, EPI_BIAS_RES = 4, EPI_LOGITS = 5, EPI_STORE_T = 6 };

// ---- loads of data another workgroup of the SAME launch has written (persistent decode kernel, persist.h) ----
// COH = true: the load must not be served by this CU's L1 (never refreshed by other CUs' stores): relaxed agent-scope
// atomic loads compile to global_load_dword(x2) ... sc1, which the XCD's L2 serves.  COH = false: plain loads (every
// launch-per-stage kernel: its inputs were written by earlier launches).
template <bool COH> __device__ inline float ldc_f32(const float* p) {
    if constexpr (!COH) return *p;
    else return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool COH> __device__ inline long long ldc_i64(const int64_t* p) {
    if constexpr (!COH) return *p;
    else return (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH> __device__ inline u32x4 ldc16(const void* p) {
    if constexpr (!COH) return ld16(p);
    else {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32x4 r; r.x = (unsigned)lo; r.y = (unsigned)(lo >> 32); r.z = (unsigned)hi; r.w = (unsigned)(hi >> 32);
        return r;
    }
}
// 16 bytes at element offset `off` of a WAVE-UNIFORM base: COH = one buffer_load_dwordx4 ... sc1 (the 8-byte atomic loads of
// ldc16 move 16 bytes as two requests; 8-byte accesses run at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md)
template <bool COH, typename E> __device__ inline u32x4 ldc16_at(const E* base, size_t off) {
    if constexpr (!COH) return ld16(base + off);
    else {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(base), 0, 0x7fffffff, 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * sizeof(E)), 0, 16);          // aux 16 = sc1
    }
}
template <bool COH> __device__ inline float4 ldc_f4_at(const float* base, size_t off) {
    const u32x4 r = ldc16_at<COH>(base, off);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
template <bool COH> __device__ inline float4 ldc_f4(const float* p) {
    const u32x4 r = ldc16<COH>(p);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
// what a tile does between requesting its weights and touching the previous stage's output: nothing in a launch-per-stage
// kernel (the launch boundary is the dependency); the persistent kernel waits for its team's arrival counter here
struct NoWait { __device__ inline void operator()() const {} };

struct NoPf { __device__ inline void operator()() const {} };

// Weight fragments of one tile (one wave's share, K fixed at compile time: KW 64-byte chunks per wave) can be requested one
// stage ahead by the persistent kernel (dec_gemm_prefetch) into ONE small register buffer shared by all stages:
// layout w0[0..KW) then (32-column tiles) w1[0..KW).
// experiment build (-DTXO_PS_W_NT=1): inside the persistent kernel the decoder weights (18 MB per position and XCD, far beyond its 4 MiB
// L2) are requested non-temporally, so that the stream does not push the latent cross attention's encoder rows (2.4 MB per XCD,
// re-read by every layer) out of that L2
#ifndef TXO_PS_W_NT
#define TXO_PS_W_NT 0
#endif
template <bool COH> __device__ inline u32x4 ld16_w(const void* p) { if constexpr (COH && TXO_PS_W_NT) return ld16_stream(p); else return ld16(p); }
constexpr int wfrag_regs(int KW, int BN) { return KW * (BN == 32 ? 2 : 1); }
constexpr int WBUF_REGS = 8;
struct WBuf { u32x4 r[WBUF_REGS]; };

template <typename T> struct DecGemmArgs {
    // problem
    int rows, N, K;                 // rows = batch, W is [N][K]
    const T* W;
    int w_tiled;                    // launch path (r05): W is the TILED copy -- the 16 rows x 64 B that one wave-instruction fetches as MFMA fragments lie
                                    // contiguously ([n / 16][k-chunk][n % 16][64 B]; rows beyond N repeat row N - 1): whole 128-byte lines per request
                                    // instead of 16 half lines (tile_weights_kernel).  0 = row-major (the persistent kernel, every other reader)
    const float* bias;              // interleaved order for the paired epilogues
    // prologue
    const T* A;                     // PRO_NONE: [rows][K]
    const float* y;                 // PRO_LN2 / PRO_LNF: [rows][K] fp32 stream
    float* x_out;                   // PRO_EMBED / PRO_LN2: residual written here (block column 0 only)
    const float* gamma; const float* beta;
    unsigned ln_lds;                // persistent kernel: LDS byte address of {gamma[K], beta[K]} copied there once, or 0 = read gamma / beta from memory
    const int64_t* tok; const float* tok_emb; const float* pos_emb;   // PRO_EMBED
    const int* t_ptr;               // decode position (device)
    int t_host;                     // >= 0: the position, known to the host (eager launches) -- saves the dependent scalar load; -1: read *t_ptr (graph replay)
    // epilogue
    float* q_out;                   // [rows][inner] fp32
    T* k_cache; T* v_cache;         // [rows*heads][tmax][64]
    int inner, heads, tmax;
    const float* resid; float* y_out; int D;    // y_out [rows][D]
    T* h_out; int F;                // [rows][F]
    T* z_cache;                     // EPI_STORE_T behind PRO_EMBED / PRO_LN2, or null: the normalised rows z (the block input) are also stored at
                                    // position t of [rows][tmax][K] -- the self attention's history in latent form (lat_attn.h)
    float* logits;                  // [rows][N]
    unsigned long long* stamps;     // diagnostic (TXO_STAMPS): per block {entry, operands landed, exit} in 10 ns ticks; null normally
    int a_tiled;                    // PRO_NONE, launch path: A is tiled like W ([m / 16][k-chunk][m % 16][64 B]) by its producer (lat_core's c output).  LAST member on
                                    // purpose: the persistent kernel's code generation follows this struct's layout, and its speed moves by 2 % with it
};

constexpr int DG_BM = 16, DG_BN = 32, DG_GROUP = 8;   // DG_GROUP: k-chunks a wave keeps in flight at once

// LN over a row spread across 16 lanes, NVMAX float4 per lane (row length 64*nv), gamma/beta in registers
template <int NVMAX, bool FAST>
__device__ inline void ln16(float4 (&v)[NVMAX], int nv, const float4 (&g)[NVMAX], const float4 (&b)[NVMAX], float inv_d) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = row16_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrt_sel<FAST>(row16_sum(q) * inv_d + LN_EPS);
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) {
        v[i].x = v[i].x * rstd * g[i].x + b[i].x; v[i].y = v[i].y * rstd * g[i].y + b[i].y;
        v[i].z = v[i].z * rstd * g[i].z + b[i].z; v[i].w = v[i].w * rstd * g[i].w + b[i].w;
    }
}

// byte offset of element k of row r in the LDS A image (rows of K*sizeof(T) bytes, 16-byte pieces swizzled)
template <typename T>
__device__ inline int a_off(int r, int k, int row_bytes, int pmask) {
    const int piece = (k * (int)sizeof(T)) >> 4;
    return r * row_bytes + ((piece ^ (r & pmask)) << 4) + ((k * (int)sizeof(T)) & 15);
}

// BN: output columns per block.  32 = two MFMA column tiles; 16 = one tile, twice the blocks and half the weight bytes
// each block has to fetch cold (FFN-out: 96 KB -> 64 KB per block).
// KW: 64-byte k-chunks per wave when K is known at compile time (K = KW * 4 chunks), 0 = any K.  With KW fixed every
// fragment array has its exact size and every loop unrolls without guards: the run-time-K form of the LN-prologue
// variants compiled to 110 branches and 324 VGPRs (one block per CU -- the 1024 blocks of the FFN-in launch at batch
// 256 ran in four rounds, 17 us); the fixed form is straight-line code at <= 128 VGPRs.
// One output tile (rows by*16.., columns bx*BN..) by 256 threads (tid 0..255; in the persistent kernel two such groups share
// a 512-thread workgroup, each with its own `smem`).  valid = false: the group has no tile in this round -- it runs the same
// barriers on clamped addresses and stores nothing.
template <typename T, int KW, int BN>
__device__ __forceinline__ void dec_gemm_prefetch(WBuf& f, const T* W, int N, int bx, int tid, bool real, int tiled = 0) {
    static_assert(KW > 0, "compile-time K only");
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK, K = KW * 4 * KCH;
    static_assert(wfrag_regs(KW, BN) <= WBUF_REGS, "prefetch buffer too small");
    // no branch around the loads (a conditional load makes hipcc's later waits vmcnt(0)): a group without a tile in that
    // stage re-reads the first rows of W, which its neighbours keep hot in L2
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const int n0 = real ? bx * BN : 0;
    const T* w0 = W + (size_t)min(n0 + lr, N - 1) * K + lg * PER16;
    const T* w1 = W + (size_t)min(n0 + 16 + lr, N - 1) * K + lg * PER16;
    int wstep = KCH;
    if (tiled) {                                              // (DecGemmArgs::w_tiled)
        const int nt = (N + 15) >> 4;
        w0 = W + ((size_t)min(n0 >> 4, nt - 1) * (K / KCH) * 16 + lr) * KCH + lg * PER16;
        w1 = W + ((size_t)min((n0 >> 4) + 1, nt - 1) * (K / KCH) * 16 + lr) * KCH + lg * PER16;
        wstep = 16 * KCH;
    }
#pragma unroll
    for (int c = 0; c < KW; ++c) {
        f.r[c] = ld16(w0 + (size_t)(wave + 4 * c) * wstep);
        if constexpr (BN == 32) f.r[KW + c] = ld16(w1 + (size_t)(wave + 4 * c) * wstep);
    }
}

// pre: the tile's weight fragments if they were requested earlier (persistent kernel), else null; pf(): called once the
// tile's own activation loads have been issued -- the persistent kernel requests the NEXT stage's weights there (vector
// memory returns in order, so anything requested before the activations would delay them)
template <typename T, int PRO, int EPI, int KW, int BN, bool COH, bool HASPRE, class Wait, class Pf>
__device__ __forceinline__ void dec_gemm_tile_pf(const DecGemmArgs<T>& a, int bx, int by, int tid, unsigned char* smem, bool valid,
                                                 Wait&& wait_prev, const WBuf& pre, Pf&& pf) {
    // value/gate-paired epilogues (GLU, GeGLU) run on ONE 16-column tile whose weight rows are interleaved by 8
    // (8 value rows, then their 8 gate rows): lane lr < 8 holds the value, lane lr + 8 the gate of output n0/2 + lr
    constexpr bool PAIRED = EPI == EPI_GLU_RES || EPI == EPI_GEGLU;
    constexpr bool TWO = BN == 32;
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK;
    constexpr bool FIXED = KW > 0;
    constexpr int GROUP = FIXED ? KW : DG_GROUP;              // k-chunks a wave keeps in flight at once
    constexpr int NVMAX = FIXED ? (KW * 4 * KCH) / 64 : 12;   // LN prologue: float4 per lane per row (row length 64 * nv)
    const int lane = tid & 63;
    // the wave id must be PROVABLY wave-uniform: MFMA ignores EXEC, so a guard the compiler lowers to EXEC
    // masking (instead of a scalar branch) would still execute the MFMA on stale registers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int m0 = by * DG_BM, n0 = bx * BN;
    const int K = FIXED ? KW * 4 * KCH : a.K, rows = a.rows;
    const int row_bytes = K * (int)sizeof(T);
    // XOR mask of the A-image swizzle: a power of two that DIVIDES the row's 16-byte piece count, so that piece ^ (r & pmask)
    // stays inside the row for every width (24 pieces at K = 192 bf16 -> mask 7; a mask of 15 there sent the tail pieces
    // into the next row's image)
    const int npiece = row_bytes >> 4;
    const int pmask = min(16, npiece & -npiece) - 1;
    unsigned long long ts0 = 0, ts1 = 0;
    if (a.stamps) ts0 = __builtin_amdgcn_s_memrealtime();

    // ---- epilogue operands: thread -> (row reg = wave, C/D lane = lane); fetched now, used at the end ----
    const int em = m0 + lg * 4 + wave;                        // output row of this thread
    const int emc = min(em, rows - 1);
    const int na = n0 + lr, nb = n0 + 16 + lr;                // the two 16-column tiles
    float e_b0 = 0.f, e_b1 = 0.f, e_res0 = 0.f, e_res1 = 0.f;
    if constexpr (PAIRED) {
        e_b0 = a.bias[na];                                    // value bias (lr < 8) or gate bias (lr >= 8) of this lane's column
        if constexpr (TWO) e_b1 = a.bias[nb];
    } else if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_LOGITS) {
        e_b0 = a.bias[min(na, a.N - 1)]; e_b1 = a.bias[min(nb, a.N - 1)];
    }

    // ---- weight fragments of the first group: issued before the prologue so they fly under it ----
    const int nch = K / KCH;                                  // 64-byte k-chunks per row
    const int my_nch = FIXED ? KW : (nch - wave + 3) >> 2;    // chunks wave, wave+4, ...
    const int wr0 = min(na, a.N - 1), wr1 = min(nb, a.N - 1);
    const T* w0 = a.W + (size_t)wr0 * K + lg * PER16;
    const T* w1 = a.W + (size_t)wr1 * K + lg * PER16;
    int wstep = KCH;                                          // elements between consecutive k-chunks of a row
    if (a.w_tiled) {                                          // (kernel-uniform) tiled copy: chunk kc of row tile nt at ((nt * nch + kc) * 16 + lr) * KCH
        const int nt = (a.N + 15) >> 4;
        w0 = a.W + ((size_t)min(na >> 4, nt - 1) * nch * 16 + lr) * KCH + lg * PER16;
        w1 = a.W + ((size_t)min(nb >> 4, nt - 1) * nch * 16 + lr) * KCH + lg * PER16;
        wstep = 16 * KCH;
    }
    u32x4 fw0[GROUP], fw1[GROUP], fa[GROUP];
    auto load_w = [&](int g0) {
#pragma unroll
        for (int c = 0; c < GROUP; ++c) if (g0 + c < my_nch) {
            const int kc = wave + 4 * (g0 + c);
            fw0[c] = ld16_w<COH>(w0 + (size_t)kc * wstep); if constexpr (TWO) fw1[c] = ld16_w<COH>(w1 + (size_t)kc * wstep);
        }
    };
    auto load_a_global = [&](int g0) {
        const int m = min(m0 + lr, rows - 1);
        const bool at = !COH && a.a_tiled;                   // (the persistent kernel's activations are row-major: its code is left as it was)
        const size_t abase = at ? ((size_t)(m >> 4) * nch * 16 + (m & 15)) * KCH : (size_t)m * K;
        const int astep = at ? 16 * KCH : KCH;
#pragma unroll
        for (int c = 0; c < GROUP; ++c) if (g0 + c < my_nch)
            fa[c] = ldc16_at<COH>(a.A, abase + (size_t)(wave + 4 * (g0 + c)) * astep + lg * PER16);
    };
    if constexpr (HASPRE) {
        static_assert(FIXED && wfrag_regs(KW, BN) <= WBUF_REGS, "prefetched fragments need a compile-time K that fits the buffer");
#pragma unroll
        for (int c = 0; c < GROUP; ++c) { fw0[c] = pre.r[c]; if constexpr (TWO) fw1[c] = pre.r[(KW + c) % WBUF_REGS]; }
    } else {
        load_w(0);
    }
    // LayerNorm affine of the prologue: static as well
    [[maybe_unused]] float4 lng[NVMAX], lnb[NVMAX];
    if constexpr (PRO != PRO_NONE) {
        const int sub = tid & 15, nv = FIXED ? NVMAX : K >> 6;
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) if (i < nv) {
            // every 16-lane row group needs the WHOLE gamma and beta: from memory that is 2 KB x 32 groups = 64 KB through the CU's
            // one vector-memory pipeline per 512-thread stage -- 0.4 us in front of the poll, whose own reads return behind them.
            // The persistent kernel keeps a copy in LDS (ds_read: another pipeline, another counter).
            if (COH && a.ln_lds) {
                lng[i] = lds_ld_f4(a.ln_lds + (i * 64 + sub * 4) * 4);
                lnb[i] = lds_ld_f4(a.ln_lds + (K + i * 64 + sub * 4) * 4);
            } else {
                lng[i] = *reinterpret_cast<const float4*>(a.gamma + i * 64 + sub * 4);
                lnb[i] = *reinterpret_cast<const float4*>(a.beta + i * 64 + sub * 4);
            }
        }
    }
    // everything above is weights / biases; everything below reads what the previous stage produced
    wait_prev();
    if constexpr (EPI == EPI_GLU_RES) {
        e_res0 = ldc_f32<COH>(a.resid + (size_t)emc * a.D + (n0 >> 1) + (lr & 7));
        if constexpr (TWO) e_res1 = ldc_f32<COH>(a.resid + (size_t)emc * a.D + (n0 >> 1) + 8 + (lr & 7));
    } else if constexpr (EPI == EPI_BIAS_RES) {
        e_res0 = ldc_f32<COH>(a.resid + (size_t)emc * a.D + na); e_res1 = ldc_f32<COH>(a.resid + (size_t)emc * a.D + nb);
    }
    int t = 0;
    if constexpr (PRO == PRO_EMBED || EPI == EPI_QKV || (EPI == EPI_STORE_T && PRO != PRO_NONE)) t = a.t_host >= 0 ? a.t_host : *a.t_ptr;
    if constexpr (PRO == PRO_NONE) { load_a_global(0); pf(); }
    // keep every fragment load ahead of the first MFMA: with K fixed this is one basic block and the machine
    // scheduler would otherwise interleave loads and MFMAs four at a time (serialising the memory latency)
    __builtin_amdgcn_sched_barrier(0);

    // ------------------------------ prologue: 16 normalised rows -> LDS ------------------------------
    if constexpr (PRO != PRO_NONE) {
        const int sub = tid & 15, r = tid >> 4, nv = FIXED ? NVMAX : K >> 6;
        const float inv_d = 1.0f / K;
        const int m = min(m0 + r, rows - 1);
        float4 v[NVMAX];
        float4 (&g)[NVMAX] = lng; float4 (&b)[NVMAX] = lnb;
        if constexpr (PRO == PRO_EMBED) {
            const float* te = a.tok_emb + (size_t)ldc_i64<COH>(a.tok + m) * K;
            const float* pe = a.pos_emb + (size_t)t * K;
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv) {
                const float4 p = *reinterpret_cast<const float4*>(te + i * 64 + sub * 4);
                const float4 q = *reinterpret_cast<const float4*>(pe + i * 64 + sub * 4);
                v[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv)
                v[i] = ldc_f4_at<COH>(a.y, (size_t)m * K + i * 64 + sub * 4);
        }
        pf();
        if constexpr (PRO == PRO_LN2) ln16<NVMAX, sizeof(T) == 2>(v, nv, g, b, inv_d);
        if constexpr (PRO == PRO_EMBED || PRO == PRO_LN2) {
            if (bx == 0 && valid && m0 + r < rows) {
#pragma unroll
                for (int i = 0; i < NVMAX; ++i) if (i < nv)
                    *reinterpret_cast<float4*>(a.x_out + (size_t)m * K + i * 64 + sub * 4) = v[i];
            }
        }
        ln16<NVMAX, sizeof(T) == 2>(v, nv, g, b, inv_d);
        // z in the storage type: the A image of this tile -- and, for the latent self attention, row t of this row's history
        // (exactly the rounded values the K/V form's k and v projections multiply: attention.py:114-127 with kv_input = x)
        [[maybe_unused]] T* zrow = nullptr;
        if constexpr (EPI == EPI_STORE_T) { if (a.z_cache && bx == 0 && valid && m0 + r < rows) zrow = a.z_cache + ((size_t)m * a.tmax + t) * K; }
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) if (i < nv) {
            unsigned char* dst = smem + a_off<T>(r, i * 64 + sub * 4, row_bytes, pmask);
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(dst) = v[i];
                if constexpr (EPI == EPI_STORE_T) { if (zrow) *reinterpret_cast<float4*>(zrow + i * 64 + sub * 4) = v[i]; }
            } else {
                union { bf16 h[4]; uint2 u; } c;
                c.h[0] = __float2bfloat16(v[i].x); c.h[1] = __float2bfloat16(v[i].y);
                c.h[2] = __float2bfloat16(v[i].z); c.h[3] = __float2bfloat16(v[i].w);
                *reinterpret_cast<uint2*>(dst) = c.u;
                if constexpr (EPI == EPI_STORE_T) { if (zrow) *reinterpret_cast<uint2*>(zrow + i * 64 + sub * 4) = c.u; }
            }
        }
        __syncthreads();
    }

    // ------------------------------ main: wave w owns k-chunks w, w+4, ... ------------------------------
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < my_nch; g0 += GROUP) {
        if (g0 > 0) { load_w(g0); if constexpr (PRO == PRO_NONE) load_a_global(g0); }
        if constexpr (PRO != PRO_NONE) {
#pragma unroll
            for (int c = 0; c < GROUP; ++c) if (g0 + c < my_nch)
                fa[c] = ld16(smem + a_off<T>(lr, (wave + 4 * (g0 + c)) * KCH + lg * PER16, row_bytes, pmask));
        }
#pragma unroll
        for (int c = 0; c < GROUP; ++c) if (g0 + c < my_nch) { mma16<T>(acc0, fa[c], fw0[c]); if constexpr (TWO) mma16<T>(acc1, fa[c], fw1[c]); }
    }

    // ------------------------------ cross-wave K reduction through LDS ------------------------------
    if (a.stamps) { asm volatile("" :: "v"(acc0[0]), "v"(acc1[0])); ts1 = __builtin_amdgcn_s_memrealtime(); }
    // partial sums go BEHIND the A image (LN-prologue tiles) so that no barrier is needed between the last fragment read and
    // the first partial-sum write; without an A image `smem` is idle here (its previous readers finished before this tile's
    // wait_prev / the caller's round barrier)
    float* red = reinterpret_cast<float*>(smem + (PRO != PRO_NONE ? DG_BM * row_bytes : 0));   // [wave][half][reg][lane]
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[r];
                                  red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[r]; }
    __syncthreads();
    // thread (wave, lane) finalises C/D register r = wave of lane `lane`: row = 4*(lane>>4) + wave, col = lane&15
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { c0 += red[((w * 2 + 0) * 4 + wave) * 64 + lane]; c1 += red[((w * 2 + 1) * 4 + wave) * 64 + lane]; }

    // ------------------------------ epilogue ------------------------------
    if (a.stamps && tid == 0) {
        unsigned long long* d = a.stamps + 3 * (size_t)(by * ((a.N + BN - 1) / BN) + bx);
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (PAIRED) {
#pragma unroll
        for (int h = 0; h < (TWO ? 2 : 1); ++h) {
            const float mine = (h ? c1 : c0) + (h ? e_b1 : e_b0);
            const float gate = dpp_mov<DPP_ROR8>(mine);       // lane lr ^ 8 holds this output's gate (rotation by 8 inside the row of 16)
            if (lr < 8 && em < rows && valid) {
                const int j = (n0 >> 1) + 8 * h + lr;         // each 16 interleaved weight rows -> 8 outputs
                if constexpr (EPI == EPI_GLU_RES) a.y_out[(size_t)em * a.D + j] = mine * sigmoid_sel<sizeof(T) == 2>(gate) + (h ? e_res1 : e_res0);
                else a.h_out[(size_t)em * a.F + j] = Elem<T>::from_f32(mine * gelu_sel<sizeof(T) == 2>(gate));
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < (TWO ? 2 : 1); ++h) {
            const int n = h ? nb : na;
            const float v = h ? c1 : c0;
            if (n >= a.N || em >= rows || !valid) continue;
            if constexpr (EPI == EPI_QKV || EPI == EPI_Q) {
                const int which = n / a.inner, f = n - which * a.inner;
                if (which == 0) a.q_out[(size_t)em * a.inner + f] = v;
                else {
                    T* cache = (which == 1) ? a.k_cache : a.v_cache;
                    cache[(((size_t)em * a.heads + (f >> 6)) * a.tmax + t) * DH + (f & 63)] = Elem<T>::from_f32(v);
                }
            } else if constexpr (EPI == EPI_BIAS_RES) {
                a.y_out[(size_t)em * a.D + n] = v + (h ? e_b1 : e_b0) + (h ? e_res1 : e_res0);
            } else if constexpr (EPI == EPI_STORE_T) {
                a.h_out[(size_t)em * a.F + n] = Elem<T>::from_f32(v);
            } else {   // EPI_LOGITS
                a.logits[(size_t)em * a.N + n] = v + (h ? e_b1 : e_b0);
            }
        }
    }
}

// A group WITHOUT a tile in a round of the persistent kernel: the workgroup barriers of dec_gemm_tile_pf and nothing else (no
// dummy tile on clamped addresses: its 16 KB of weight requests would sit in the CU's vector-memory pipeline in front of the
// other group's)
template <int PRO, class Wait>
__device__ __forceinline__ void dec_gemm_idle(Wait&& wait_prev) {
    wait_prev();
    if constexpr (PRO != PRO_NONE) __syncthreads();           // A image written
    __syncthreads();                                          // partial sums written
}

template <typename T, int PRO, int EPI, int KW, int BN, bool COH, class Wait>
__device__ __forceinline__ void dec_gemm_tile(const DecGemmArgs<T>& a, int bx, int by, int tid, unsigned char* smem, bool valid,
                                              Wait&& wait_prev) {
    const WBuf none{};
    dec_gemm_tile_pf<T, PRO, EPI, KW, BN, COH, false>(a, bx, by, tid, smem, valid, wait_prev, none, NoPf{});
}

template <typename T, int PRO, int EPI, int KW, int BN = DG_BN>
__global__ __launch_bounds__(256, (KW > 0 && KW <= 8 && (PRO == PRO_NONE || KW * 4 * Elem<T>::KCHUNK <= 256) ? 4 : 2))
void dec_gemm_kernel(DecGemmArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    dec_gemm_tile<T, PRO, EPI, KW, BN, false>(a, blockIdx.x, blockIdx.y, threadIdx.x, smem, true, NoWait{});
}

// ---- >= 128 rows of a wide decoder (launch path only) ---------------------------------------------------------------------------
// dec_gemm_tile_pf's PRO_NONE form with RT row tiles of 16 per block that SHARE the block's weight fragments: 16-row blocks
// re-read every weight slice rows/16 times from L2 (74 MB per out-projection at 256 rows x 768), which is what bounds those
// launches; 64-row blocks read a quarter of that.  Same K split over the four waves, same reduction order, same epilogue
// arithmetic as the 16-row tile -> a row's bits do not depend on which of the two kernels computed it.  A separate function on
// purpose: the 16-row tile is inlined ~20 times into the persistent decode kernel, whose code generation is sensitive to any
// change of it (a templated-in RT = 1 cost that kernel 3 %).
template <typename T, int EPI, int KW, int BN, int RT>
__device__ __forceinline__ void dec_gemm_wide_tile(const DecGemmArgs<T>& a, int bx, int by, int tid, unsigned char* smem) {
    constexpr bool PAIRED = EPI == EPI_GLU_RES || EPI == EPI_GEGLU;
    constexpr bool TWO = BN == 32;
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK;
    static_assert(KW > 0, "compile-time K only");
    constexpr int GROUP = KW > 16 ? DG_GROUP : KW;            // k-chunks a wave keeps in flight at once
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int m0 = by * DG_BM * RT, n0 = bx * BN;
    constexpr int K = KW * 4 * KCH;
    const int rows = a.rows;
    const int em0 = m0 + lg * 4 + wave;                       // output row of this thread in row tile 0 (row tile rt: + 16 rt)
    const int na = n0 + lr, nb = n0 + 16 + lr;
    float e_b0 = 0.f, e_b1 = 0.f, e_res0[RT], e_res1[RT];
    if constexpr (PAIRED) { e_b0 = a.bias[na]; if constexpr (TWO) e_b1 = a.bias[nb]; }
    else if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_LOGITS) { e_b0 = a.bias[min(na, a.N - 1)]; e_b1 = a.bias[min(nb, a.N - 1)]; }
    const T* w0 = a.W + (size_t)min(na, a.N - 1) * K + lg * PER16;
    const T* w1 = a.W + (size_t)min(nb, a.N - 1) * K + lg * PER16;
    int wstep = KCH;
    if (a.w_tiled) {                                          // (dec_gemm_tile_pf)
        const int nt = (a.N + 15) >> 4;
        w0 = a.W + ((size_t)min(na >> 4, nt - 1) * (K / KCH) * 16 + lr) * KCH + lg * PER16;
        w1 = a.W + ((size_t)min(nb >> 4, nt - 1) * (K / KCH) * 16 + lr) * KCH + lg * PER16;
        wstep = 16 * KCH;
    }
    u32x4 fw0[GROUP], fw1[GROUP], fa[RT][GROUP];
    auto load_group = [&](int g0) {
#pragma unroll
        for (int c = 0; c < GROUP; ++c) {
            const int kc = wave + 4 * (g0 + c);
            fw0[c] = ld16(w0 + (size_t)kc * wstep); if constexpr (TWO) fw1[c] = ld16(w1 + (size_t)kc * wstep);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = min(m0 + rt * DG_BM + lr, rows - 1);
#pragma unroll
            for (int c = 0; c < GROUP; ++c)
                fa[rt][c] = ld16(a.A + (a.a_tiled ? ((size_t)(m >> 4) * (K / KCH) * 16 + (m & 15)) * KCH + (size_t)(wave + 4 * (g0 + c)) * (16 * KCH)
                                                  : (size_t)m * K + (wave + 4 * (g0 + c)) * KCH) + lg * PER16);
        }
    };
    load_group(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int emc = min(em0 + rt * DG_BM, rows - 1);
        e_res0[rt] = 0.f; e_res1[rt] = 0.f;
        if constexpr (EPI == EPI_GLU_RES) {
            e_res0[rt] = a.resid[(size_t)emc * a.D + (n0 >> 1) + (lr & 7)];
            if constexpr (TWO) e_res1[rt] = a.resid[(size_t)emc * a.D + (n0 >> 1) + 8 + (lr & 7)];
        } else if constexpr (EPI == EPI_BIAS_RES) {
            e_res0[rt] = a.resid[(size_t)emc * a.D + min(na, a.N - 1)]; e_res1[rt] = a.resid[(size_t)emc * a.D + min(nb, a.N - 1)];
        }
    }
    int t = 0;
    if constexpr (EPI == EPI_QKV) t = a.t_host >= 0 ? a.t_host : *a.t_ptr;
    __builtin_amdgcn_sched_barrier(0);                        // every fragment load ahead of the first MFMA (see dec_gemm_tile_pf)
    f32x4 acc0[RT], acc1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { acc0[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int g0 = 0; g0 < KW; g0 += GROUP) {
        if (g0 > 0) load_group(g0);
#pragma unroll
        for (int c = 0; c < GROUP; ++c)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { mma16<T>(acc0[rt], fa[rt][c], fw0[c]); if constexpr (TWO) mma16<T>(acc1[rt], fa[rt][c], fw1[c]); }
    }
    float* red = reinterpret_cast<float*>(smem);              // [rt][wave][half][reg][lane]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[rt * 2048 + ((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[rt][r];
                                      red[rt * 2048 + ((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[rt][r]; }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { c0 += red[rt * 2048 + ((w * 2 + 0) * 4 + wave) * 64 + lane]; c1 += red[rt * 2048 + ((w * 2 + 1) * 4 + wave) * 64 + lane]; }
        const int em = em0 + rt * DG_BM;
        if constexpr (PAIRED) {
#pragma unroll
            for (int h = 0; h < (TWO ? 2 : 1); ++h) {
                const float mine = (h ? c1 : c0) + (h ? e_b1 : e_b0);
                const float gate = dpp_mov<DPP_ROR8>(mine);
                if (lr < 8 && em < rows) {
                    const int j = (n0 >> 1) + 8 * h + lr;
                    if constexpr (EPI == EPI_GLU_RES) a.y_out[(size_t)em * a.D + j] = mine * sigmoid_sel<sizeof(T) == 2>(gate) + (h ? e_res1[rt] : e_res0[rt]);
                    else a.h_out[(size_t)em * a.F + j] = Elem<T>::from_f32(mine * gelu_sel<sizeof(T) == 2>(gate));
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < (TWO ? 2 : 1); ++h) {
                const int n = h ? nb : na;
                const float v = h ? c1 : c0;
                if (n >= a.N || em >= rows) continue;
                if constexpr (EPI == EPI_QKV) {
                    const int which = n / a.inner, f = n - which * a.inner;
                    if (which == 0) a.q_out[(size_t)em * a.inner + f] = v;
                    else {
                        T* cache = (which == 1) ? a.k_cache : a.v_cache;
                        cache[(((size_t)em * a.heads + (f >> 6)) * a.tmax + t) * DH + (f & 63)] = Elem<T>::from_f32(v);
                    }
                } else if constexpr (EPI == EPI_BIAS_RES) {
                    a.y_out[(size_t)em * a.D + n] = v + (h ? e_b1 : e_b0) + (h ? e_res1[rt] : e_res0[rt]);
                } else {
                    a.logits[(size_t)em * a.N + n] = v + (h ? e_b1 : e_b0);
                }
            }
        }
    }
}

template <typename T, int EPI, int KW, int BN, int RT>
__global__ __launch_bounds__(256, 1) void dec_gemm_wide_kernel(DecGemmArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    dec_gemm_wide_tile<T, EPI, KW, BN, RT>(a, blockIdx.x, blockIdx.y, threadIdx.x, smem);
}

// W [N][K] row-major -> the tiled copy (DecGemmArgs::w_tiled); one thread per 16-byte piece
template <typename T>
__global__ void tile_weights_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int K) {
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK;
    const int nch = K / KCH;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)((N + 15) / 16) * nch * 64;
    if (idx >= total) return;
    const int lg = idx & 3, lr = (idx >> 2) & 15;
    const size_t blk = idx >> 6;
    const int kc = (int)(blk % nch), nt = (int)(blk / nch);
    st16(dst + idx * PER16, ld16(src + (size_t)min(nt * 16 + lr, N - 1) * K + kc * KCH + lg * PER16));
}

template <typename T>
inline size_t dec_gemm_lds_bytes(int K, bool has_pro) {
    const size_t red = (size_t)4 * 2 * 4 * 64 * 4;
    const size_t img = has_pro ? (size_t)DG_BM * K * sizeof(T) : 0;
    return img + red;                                         // the partial sums sit behind the A image
}

}  // namespace txo
