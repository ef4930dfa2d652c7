// Encoder self-attention (non-causal), flash-style, exact-f32 MFMA.
// Reference: MultiHeadAttention.forward, model/attention.py:148-173 with causal=False and all-True masks:
//   energy = q k^T * 0.125 ; softmax over keys ; out = attn v ; merge heads 'b h n d -> b n (h d)'.
// The reference materialises (and clones twice) the [B,h,N,N] score tensor; here scores never leave
// registers.
//
// Inputs q,k,v: head-major [B*heads][N][64] f32 (written by the QKV GEMM epilogue).  Output: [B*N][heads*64].
//
// gfx950 mapping.  Block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries (two
// 16-wide MFMA column tiles).  K/V stream through LDS in 64-key stages (double buffered, 64 KB).
// S^T = K Q^T is computed with the key on the MFMA row and the query on the lane column, so that
//  (1) the softmax statistics of a query live in one lane column (registers + 2 cross-group shuffles),
//  (2) the P tile in its accumulator layout IS the B operand of O^T += V^T P^T (16x16x4: lane group g,
//      register r holds key 4g+r; the V fragment is simply read for that same key) -- no LDS round trip
//      and no lane movement for P.
// MFMA: v_mfma_f32_16x16x4_f32, 64 per wave per 16 keys (32 for QK^T, 32 for PV) = the f32 matrix peak
// rate; everything else (8 exps, 32 rescale FMAs, 20 LDS reads per 16 keys) hides behind it.
// Bound: MFMA f32 (157 TF peak).
#pragma once
#include <type_traits>
#include "common.h"

namespace txo {

constexpr int EA_QBLK = 128, EA_KSTAGE = 64;

__device__ inline int swz256(int row, int piece) { return row * 256 + ((piece ^ (row & 15)) << 4); }

// Block -> (batch*head, query block).  One-dimensional grid of 8 * ceil(nbh / 8) * nq workgroups: workgroups b and b + 8 run on one XCD
// (round-robin dispatch), so the nq query blocks of ONE (image, head) go to ONE XCD and its L2 fetches that head's K / V panel once for
// all of them.  (With the (query block, batch*head) grid the five query blocks of a head landed on five XCDs: every K / V panel crossed
// the fabric five times -- 2.3 GB per ViT-Base layer in 0.44 ms, i.e. the attention kernel was at the fabric's rate, not the VALU's.)
// rev: the (image, head) pairs in descending order (engine.hip: the encoder's kernels walk the rows alternately up and down, so that
// each starts on what the previous one wrote last)
__device__ inline bool ea_block(int nq, int nbh, int& bh, int& qb, int rev = 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bh = (slot / nq) * 8 + xcd; qb = slot - (slot / nq) * nq;
    if (bh >= nbh) return false;
    if (rev) bh = nbh - 1 - bh;
    return true;
}
inline dim3 ea_grid(int nq, int nbh) { return dim3(((nbh + 7) / 8) * 8 * nq); }

template <typename TO>
__global__ __launch_bounds__(256) void enc_attn_kernel(const float* __restrict__ Q, const float* __restrict__ Kg,
                                                       const float* __restrict__ Vg, TO* __restrict__ out, int N,
                                                       int heads, int nbh) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][EA_KSTAGE * 256];   // [buf][K|V]
    int bh, qblk;
    if (!ea_block((N + EA_QBLK - 1) / EA_QBLK, nbh, bh, qblk)) return;
    const int b = bh / heads, head = bh - b * heads;
    const int q0 = qblk * EA_QBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lg = lane >> 4;
    const float* Qb = Q + (size_t)bh * N * DH;
    const float* Kb = Kg + (size_t)bh * N * DH;
    const float* Vb = Vg + (size_t)bh * N * DH;

    // Q fragments (B operand of S^T = K Q^T): lane (query = lc, group lg) holds Q[query][16kc + 4lg + e]
    u32x4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + wave * 32 + qt * 16 + lc, N - 1);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            float4 t = *reinterpret_cast<const float4*>(Qb + (size_t)qrow * DH + kc * 16 + lg * 4);
            t.x *= ATTN_SCALE; t.y *= ATTN_SCALE; t.z *= ATTN_SCALE; t.w *= ATTN_SCALE;   // exact (power of two)
            qf[qt][kc] = __builtin_bit_cast(u32x4, t);
        }
    }

    f32x4 o[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    // staging: 64 keys x 256 B = 1024 16-byte pieces per operand, 4 per thread
    u32x4 rk[4], rv[4];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, piece = idx & 15;
            const int key = min(s * EA_KSTAGE + row, N - 1);
            rk[i] = ld16(Kb + (size_t)key * DH + piece * 4);
            rv[i] = ld16(Vb + (size_t)key * DH + piece * 4);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, piece = idx & 15;
            st16(&lds[buf][0][swz256(row, piece)], rk[i]);
            st16(&lds[buf][1][swz256(row, piece)], rv[i]);
        }
    };

    const int nstage = (N + EA_KSTAGE - 1) / EA_KSTAGE;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) load_stage(s + 1);
        const unsigned char* Ks = lds[buf][0];
        const unsigned char* Vs = lds[buf][1];

        // ---- S^T for 64 keys x 32 queries ----
        f32x4 sc[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 kf[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) kf[kc] = ld16(Ks + swz256(kt * 16 + lc, kc * 4 + lg));
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) mma16<float>(a, kf[kc], qf[qt][kc]);
                sc[qt][kt] = a;
            }
        }
        // keys past N only exist in the last stage
        const int kbase = s * EA_KSTAGE;
        if (kbase + EA_KSTAGE > N) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + kt * 16 + lg * 4 + r >= N) { sc[0][kt][r] = -1e30f; sc[1][kt][r] = -1e30f; }
        }
        // ---- online softmax (per query = per lane column; keys spread over regs and lane groups) ----
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = sc[qt][0][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[qt][kt][r]);
            mx = grp4_max(mx);
            const float m_new = fmaxf(m_run[qt], mx);
            const float alpha = expf(m_run[qt] - m_new);
            m_run[qt] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float p = expf(sc[qt][kt][r] - m_new); sc[qt][kt][r] = p; ps += p; }
            l_run[qt] = l_run[qt] * alpha + ps;          // per-lane partial; groups are summed at the end
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] *= alpha;
        }
        // ---- O^T += V^T P^T : A = V[key 16kt+4lg+r][d = 16dt+lc] (one float per lane), B = P regs ----
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kt * 16 + lg * 4 + r;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int col = dt * 16 + lc;
                    const float vv = *reinterpret_cast<const float*>(Vs + swz256(row, col >> 2) + (col & 3) * 4);
                    o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sc[0][kt][r], o[0][dt], 0, 0, 0);
                    o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sc[1][kt][r], o[1][dt], 0, 0, 0);
                }
            }
        }
        if (s + 1 < nstage) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise, transpose through LDS (wave-private 32 x 64 f32 tile), store whole 256-byte rows ----
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]) + wave * (32 * 64);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = 1.0f / grp4_sum(l_run[qt]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(qt * 16 + lc) * 64 + ((dt * 16 + lg * 4 + r) ^ ((lc & 7) << 2))] = o[qt][dt][r] * inv;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave's own LDS writes are done (wave-private tile)
    const int inner = heads * DH;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, qq = idx >> 4, piece = idx & 15;
        const int qrow = q0 + wave * 32 + qq;
        if (qrow < N) {
            const int c0 = (piece * 4) ^ ((qq & 7) << 2);
            const float4 v4 = *reinterpret_cast<const float4*>(&tile[qq * 64 + c0]);
            TO* dst = out + ((size_t)(b * N + qrow)) * inner + head * DH + piece * 4;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(dst) = v4;
            } else {
                union { bf16 h[4]; uint2 u; } t;
                t.h[0] = __float2bfloat16(v4.x); t.h[1] = __float2bfloat16(v4.y);
                t.h[2] = __float2bfloat16(v4.z); t.h[3] = __float2bfloat16(v4.w);
                *reinterpret_cast<uint2*>(dst) = t.u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 MFMA variant (perf mode): same structure, v_mfma_f32_16x16x32_bf16 (16x the f32 matrix rate).
//   S^T = K Q^T : A = K rows from LDS (128-byte rows, XOR-swizzled), B = Q fragments in registers (pre-scaled).
//   P   -> bf16 : the S^T accumulators of two 16-key tiles, converted and packed, ARE the B operand of one 32-key
//                 k-step of O^T += V^T P^T (k-slot j of lane group g = key 4g+j of the first tile for j < 4,
//                 key 16+4g+(j-4) of the second) -- no LDS round trip for P.
// (The first form of it, with V transposed by 2-byte scatters on its way into LDS, was removed in r06: git history, r01-r05.)

// two f32 -> packed bf16 pair (lo = a, hi = b) in one v_cvt_pk_bf16_f32
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline unsigned pack_bf16x2(float a, float b) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, h);
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float EAB_GROW = 6.0f;      // base-2 units: P stays below 64 between shifts (bf16 / f32 keep their relative precision)


// ---------------------------------------------------------------------------------------------------------
// bf16 variant 2 (default in perf mode): the same tiling and the same S^T = K Q^T / O^T += V^T P^T orientation, with the
// work per score cut to what the matrix pipe cannot do:
//   * V stays row-major in LDS ([key][64 d], 16-byte pieces XOR-swizzled with (key & 6)); the A operand of O^T += V^T P^T is
//     read with ds_read_b64_tr_b16, gfx950's transposing LDS read (a 16-lane group fetches 4 keys x 16 d and every lane gets
//     ITS d for those 4 keys = the k-slots the P accumulator layout dictates).  The 2-byte scatter that transposed V on its
//     way into LDS (16 ds_write_b16 per thread and stage) is gone;
//   * the running maximum enters the QK^T MFMA as its initial accumulator (C = -m): scores come out already shifted, and as
//     long as no query's maximum grows in a stage (a wave-uniform test; true for most stages once the first few have been
//     seen) nothing is subtracted and the output accumulators are not rescaled;
//   * the softmax normaliser is accumulated by the matrix pipe too: a fifth A tile of ones makes sum_k P[k] one more
//     accumulator tile of O^T (4 extra MFMAs per stage instead of 32 adds per lane, and it is the sum of exactly the bf16 P
//     values that multiply V).
// Per 64-key stage and wave: 36 MFMAs, 32 v_exp_f32, 16 packed conversions, ~16 three-input maxima.
// q,k,v: bf16 head-major [B*heads][N][64].  Bound: MFMA bf16 / softmax VALU, about equal.
template <typename TO>
__global__ __launch_bounds__(256) void enc_attn_bf16_v2_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kg,
                                                               const bf16* __restrict__ Vg, TO* __restrict__ out, int N, int heads, int nbh, int rev) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][EA_KSTAGE * 128];   // [buf][K | V], 128-byte rows
    static_assert(sizeof(lds) >= 4 * 32 * 64 * 4, "epilogue tile must fit");
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    int bh, qblk;
    if (!ea_block((N + EA_QBLK - 1) / EA_QBLK, nbh, bh, qblk, rev)) return;
    const int b = bh / heads, head = bh - b * heads;
    const int q0 = qblk * EA_QBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lg = lane >> 4;
    const bf16* Qb = Q + (size_t)bh * N * DH;
    const bf16* Kb = Kg + (size_t)bh * N * DH;
    const bf16* Vb = Vg + (size_t)bh * N * DH;

    // Q fragments, pre-scaled by 0.125 * log2(e) (scores in base-2 units: one v_exp_f32 per score)
    u32x4 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + wave * 32 + qt * 16 + lc, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 raw = ld16_once(Qb + (size_t)qrow * DH + ks * 32 + lg * 8);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2(__uint_as_float(w[k] << 16) * (ATTN_SCALE * LOG2E), __uint_as_float(w[k] & 0xffff0000u) * (ATTN_SCALE * LOG2E));
            qf[qt][ks] = u32x4{o[0], o[1], o[2], o[3]};
        }
    }

    f32x4 o[2][5];                                           // [query tile][d tile 0..3 | normaliser]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {0.f, 0.f};                             // base of the shifted scores; set by the first stage

    // staging: 64 keys x 128 B = 512 16-byte pieces per operand, 2 per thread
    u32x4 rk[2], rv[2];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, piece = idx & 7;
            const int key = min(s * EA_KSTAGE + row, N - 1);
            rk[i] = ld16_once(Kb + (size_t)key * DH + piece * 8);
            rv[i] = ld16_once(Vb + (size_t)key * DH + piece * 8);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, piece = idx & 7;
            st16(&lds[buf][0][swz128(row, piece)], rk[i]);
            st16(&lds[buf][1][row * 128 + ((piece ^ (row & 6)) << 4)], rv[i]);
        }
    };
    // transposed V reads: lane (group lg, q = lc >> 2, p = lc & 3) addresses key 4 lg + q (+16 / +32 per instruction), d = 16 dt + 4p..
    const int vq = lc >> 2, vp = lc & 3, vrow = lg * 4 + vq, vx = vrow & 6;
    const int vbase = vrow * 128 + ((vp >> 1) << 4) + ((vp & 1) << 3);
    int vcol[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vcol[dt] = ((2 * dt) ^ vx) << 4;
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    const int nstage = (N + EA_KSTAGE - 1) / EA_KSTAGE;
    // r05: the ragged ends.  N = 589 is 9 key stages of 64 + one of 13, and 4 query blocks of 128 + one of 77: the last stage runs only the
    // 16-key tiles / 32-key k-steps that hold a key (one of four / one of two), and a wave whose 32 queries all lie beyond N only stages
    // K / V and keeps the barriers.  Same arithmetic for every key and query that exists (a skipped tile contributed exact zeros).
    const bool active = __builtin_amdgcn_readfirstlane(q0 + wave * 32) < N;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    auto stage = [&](int s, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;        // LAST: the stage may be ragged (fewer than 64 keys)
        const int buf = s & 1;
        if (s + 1 < nstage) load_stage(s + 1);
        if (active) {
        const unsigned char* Ks = lds[buf][0];
        const unsigned char* Vs = lds[buf][1];
        const int kbase = s * EA_KSTAGE;
        const int nkt = LAST ? min(4, (N - kbase + 15) >> 4) : 4;      // 16-key tiles of this stage that hold a key (block-uniform)

        // ---- S^T - m for 64 keys x 32 queries: 16 MFMA, the running maximum as the initial accumulator ----
        f32x4 sc[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (LAST && kt >= nkt) {
                asm volatile("; key tile past the last key" ::: "memory");   // a real (scalar) branch around the tile's MFMAs
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sc[qt][kt] = f32x4{-1e30f, -1e30f, -1e30f, -1e30f};
                continue;
            }
            u32x4 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = ld16(Ks + swz128(kt * 16 + lc, ks * 4 + lg));
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float c = -m_run[qt];
                f32x4 a = {c, c, c, c};
                mma16<bf16>(a, kf[0], qf[qt][0]);
                mma16<bf16>(a, kf[1], qf[qt][1]);
                sc[qt][kt] = a;
            }
        }
        if (LAST && kbase + EA_KSTAGE > N) {
            asm volatile("; ragged last stage" ::: "memory");    // keep this a branch: if-converted it costs ~60 VALU ops in EVERY stage
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + kt * 16 + lg * 4 + r >= N) { sc[0][kt][r] = -1e30f; sc[1][kt][r] = -1e30f; }
        }
        // ---- stage maximum relative to the running one; rescale only when some query's maximum grows ----
        float rel[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = fmaxf(fmaxf(sc[qt][0][0], sc[qt][0][1]), fmaxf(sc[qt][0][2], sc[qt][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) mx = fmaxf(fmaxf(mx, sc[qt][kt][0]), fmaxf(fmaxf(sc[qt][kt][1], sc[qt][kt][2]), sc[qt][kt][3]));
            rel[qt] = grp4_max(mx);
        }
        const bool first = s == 0;
        // The shift is only an overflow guard (softmax is invariant to it): it follows the maximum only when some query's
        // maximum has grown by more than 2^EAB_GROW since the last shift.  Growth by any amount would fire in nearly every
        // stage (with 32 queries per wave SOME maximum grows in 97 % of the stages of a 589-key row).
        if (first || __builtin_amdgcn_ballot_w64(rel[0] > EAB_GROW || rel[1] > EAB_GROW) != 0ull) {
            asm volatile("; a running maximum grows: shift the scores, rescale the accumulators" ::: "memory");   // a real branch (see above)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float delta = first ? rel[qt] : fmaxf(rel[qt], 0.f);
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
                m_run[qt] += delta;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[qt][kt][r] -= delta;
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) o[qt][dt] *= alpha;
            }
        }
        // ---- P = 2^score, packed to the bf16 k-step operands ----
        u32x4 pb[2][2];                                        // [qt][32-key k-step]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            unsigned pk[8];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (LAST && kt >= nkt) { pk[2 * kt] = 0u; pk[2 * kt + 1] = 0u; continue; }     // (2^-1e30 = 0: the same zeros, not computed)
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(sc[qt][kt][r]);
                pk[2 * kt] = pack_bf16x2(p[0], p[1]); pk[2 * kt + 1] = pack_bf16x2(p[2], p[3]);
            }
            pb[qt][0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
            pb[qt][1] = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
        // ---- O^T += V^T P^T (16 MFMA) and the normaliser tile (4 MFMA) ----
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if (LAST && k2 * 2 >= nkt) {
                asm volatile("; k-step past the last key" ::: "memory");
                continue;                                      // its P operand is all zeros
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* va = Vs + k2 * (32 * 128) + vbase + vcol[dt];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uint32_t)(uintptr_t)va);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uint32_t)(uintptr_t)(va + 16 * 128));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                const u32x4 vf = {l2.x, l2.y, h2.x, h2.y};
                mma16<bf16>(o[0][dt], vf, pb[0][k2]);
                mma16<bf16>(o[1][dt], vf, pb[1][k2]);
            }
            mma16<bf16>(o[0][4], ones, pb[0][k2]);
            mma16<bf16>(o[1][4], ones, pb[1][k2]);
        }
        }
        if (s + 1 < nstage) store_stage(buf ^ 1);
        __syncthreads();
    };
    for (int s = 0; s + 1 < nstage; ++s) stage(s, std::false_type{});
    stage(nstage - 1, std::true_type{});
    if (!active) return;                                       // (behind the last barrier)

    // ---- normalise, transpose through LDS (wave-private 32 x 64 f32 tile), store whole rows ----
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]) + wave * (32 * 64);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = 1.0f / o[qt][4][0];                 // every row of the normaliser tile holds sum_k P[k] of this lane's query
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(qt * 16 + lc) * 64 + ((dt * 16 + lg * 4 + r) ^ ((lc & 7) << 2))] = o[qt][dt][r] * inv;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int inner = heads * DH;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, qq = idx >> 4, piece = idx & 15;
        const int qrow = q0 + wave * 32 + qq;
        if (qrow < N) {
            const int c0 = (piece * 4) ^ ((qq & 7) << 2);
            const float4 v4 = *reinterpret_cast<const float4*>(&tile[qq * 64 + c0]);
            TO* dst = out + ((size_t)(b * N + qrow)) * inner + head * DH + piece * 4;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(dst) = v4;
            } else {
                union { bf16 h[4]; uint2 u; } t;
                t.h[0] = __float2bfloat16(v4.x); t.h[1] = __float2bfloat16(v4.y);
                t.h[2] = __float2bfloat16(v4.z); t.h[3] = __float2bfloat16(v4.w);
                *reinterpret_cast<uint2*>(dst) = t.u;
            }
        }
    }
}

}  // namespace txo
