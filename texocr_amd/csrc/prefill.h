// Multi-position ("prefill") decoder forward: Transformer.forward (reference model/decoder.py:41-67) over ALL t positions of a
// prefix in one pass -- token + position embedding, the decoder stack with a CAUSAL self attention (attention.py:158-163) and the
// cross attention over the cached encoder projections, final LayerNorm, logits -- on encoder-style GEMMs over B*t rows
// (gemm_big.h / gemm_pp.h with the decoder's weights) instead of t single-position steps.  It also FILLS the self-attention
// K/V cache rows 0..t-1, so txo_decode_step(t, ...) continues behind it.  Used by decoder.net() (teacher-forced logits) and by
// the sliding window (decoder.py:99-100: once the output is longer than the positional table every further token re-runs its
// window -- one prefill per token instead of max_len steps).
//
// Kernels here: the embedding rows, the q/k/v scatter epilogue (q head-major for the attention kernel, k/v straight into the
// cache layout), and a flash attention with separate query / key counts and an optional causal mask on exact-f32 MFMA
// (the structure of enc_attn_kernel; K/V of either storage type are widened to f32 on their way into LDS).
// Bound: MFMA f32 (157 TF) for the attention, MFMA for the GEMMs; this path is not the benchmark's hot loop.
#pragma once
#include "common.h"
#include "enc_attn.h"
#include "gemm_big.h"

namespace txo {

// x[b*t + p][:] = tok_emb[tokens[b][p]] + pos_emb[p]      (decoder.py:51-52, attention.py:30-32)
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ tokens, int tok_stride, const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb, float* __restrict__ x, int rows, int t, int D, int V) {
    const int per = D >> 2;                                   // float4 per row
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * per) return;
    const int m = (int)(i / per), c = (int)(i - (size_t)m * per) * 4;
    const int b = m / t, p = m - b * t;
    long long id = tokens[(size_t)b * tok_stride + p];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);                 // a token outside the vocabulary must not read outside the table
    const float4 e = *reinterpret_cast<const float4*>(tok_emb + (size_t)id * D + c);
    const float4 q = *reinterpret_cast<const float4*>(pos_emb + (size_t)p * D + c);
    *reinterpret_cast<float4*>(x + (size_t)m * D + c) = make_float4(e.x + q.x, e.y + q.y, e.z + q.z, e.w + q.w);
}

// q,k,v scatter of the prefill's QKV GEMM: n = (which, head, d); row m = (image, position).  q -> [image*heads][t][64] of T
// (the attention kernel's query layout), k / v -> the self-attention cache [image*heads][tmax][64] at row `position`.
template <typename T> struct EpiHeadsKV {
    T* q; T* kc; T* vc; int inner, heads, t, tmax;
    static constexpr bool PAIRED = false;
    static constexpr int ST = store8_insts<T>();        // 16-byte store instructions per fin() (gemm_pp.h)
    static constexpr int NCB = 0;
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        const int which = n / inner, f = n - which * inner, head = f >> 6, d = f & 63;   // 8 columns never straddle a head
        const int b = m / t, p = m - b * t;
        T* dst = which == 0 ? q + (((size_t)b * heads + head) * t + p) * DH + d
                            : (which == 1 ? kc : vc) + (((size_t)b * heads + head) * tmax + p) * DH + d;
        store8<T>(dst, v);
    }
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int, float (&)[32]) const {}
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&)[32], const float (&)[10], bool valid) const {
        if (valid) (*this)(m, n, v);
    }
};

// 4 consecutive elements of a row as f32 (16 bytes of f32, or 8 bytes of bf16 widened)
template <typename TI> __device__ inline u32x4 ld4_f32(const TI* p);
template <> __device__ inline u32x4 ld4_f32<float>(const float* p) { return ld16(p); }
template <> __device__ inline u32x4 ld4_f32<bf16>(const bf16* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    u32x4 o; o.x = r.x << 16; o.y = r.x & 0xffff0000u; o.z = r.y << 16; o.w = r.y & 0xffff0000u;
    return o;
}

// softmax(q k^T / 8 [+ causal mask]) v for nq queries against nk keys of one (image, head):
//   Q [bh][nq][64], K / V [bh][kv_rows][64] (only the first nk rows are read), out [(image*nq + query)][heads*64].
//   CAUSAL: key j takes part for query i iff j <= i + (nk - nq)  (attention.py:158-163 with F.pad(mask, (j - i, 0))).
// Structure of enc_attn_kernel: block = 128 queries, K/V in 64-key f32 LDS stages, S^T = K Q^T so that softmax statistics are
// per lane and the P accumulators are the B operand of O^T += V^T P^T.
// KMASK: the padding mask of decoder.net / decoder.generate (attention.py:130-155: energy filled with -FLT_MAX where q_mask (x) k_mask is
//   False).  kmask [images][kmask_stride] bytes, 0 = padding.  A key that is padding is never attended by a query that is not; the row of a
//   query that IS padding is unspecified here as on the single-position path (the reference softmaxes it uniformly over all keys, future ones
//   included; nothing reads it): it ignores the mask, so that it stays finite -- its k / v rows land in the cache and are masked at every use.
template <typename TI, typename TO, bool CAUSAL, bool KMASK = false>
__global__ __launch_bounds__(256) void attn_mq_kernel(const TI* __restrict__ Q, const TI* __restrict__ Kg, const TI* __restrict__ Vg,
                                                      TO* __restrict__ out, int nq, int nk, int kv_rows, int heads,
                                                      const unsigned char* __restrict__ kmask = nullptr, int kmask_stride = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][EA_KSTAGE * 256];   // [buf][K|V], f32 rows
    const int bh = blockIdx.y, b = bh / heads, head = bh - b * heads;
    const int q0 = blockIdx.x * EA_QBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lg = lane >> 4;
    const TI* Qb = Q + (size_t)bh * nq * DH;
    const TI* Kb = Kg + (size_t)bh * kv_rows * DH;
    const TI* Vb = Vg + (size_t)bh * kv_rows * DH;
    const int off = nk - nq;

    u32x4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + wave * 32 + qt * 16 + lc, nq - 1);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            float4 tq = __builtin_bit_cast(float4, ld4_f32<TI>(Qb + (size_t)qrow * DH + kc * 16 + lg * 4));
            tq.x *= ATTN_SCALE; tq.y *= ATTN_SCALE; tq.z *= ATTN_SCALE; tq.w *= ATTN_SCALE;   // exact (power of two)
            qf[qt][kc] = __builtin_bit_cast(u32x4, tq);
        }
    }
    f32x4 o[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    [[maybe_unused]] bool q_valid[2] = {true, true};          // KMASK: this lane's query is not padding (query i sits at position i + off)
    if constexpr (KMASK) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
            q_valid[qt] = kmask[(size_t)b * kmask_stride + min(q0 + wave * 32 + qt * 16 + lc, nq - 1) + off] != 0;
    }

    u32x4 rk[4], rv[4];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, piece = idx & 15;
            const int key = min(s * EA_KSTAGE + row, nk - 1);
            rk[i] = ld4_f32<TI>(Kb + (size_t)key * DH + piece * 4);
            rv[i] = ld4_f32<TI>(Vb + (size_t)key * DH + piece * 4);
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
    // causal: no key beyond the block's last query takes part (block-uniform stage count)
    const int last_key = CAUSAL ? min(nk - 1, min(q0 + EA_QBLK - 1, nq - 1) + off) : nk - 1;
    const int nstage = last_key / EA_KSTAGE + 1;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) load_stage(s + 1);
        const unsigned char* Ks = lds[buf][0];
        const unsigned char* Vs = lds[buf][1];
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
        const int kbase = s * EA_KSTAGE;
        if constexpr (KMASK) {                                // padded keys, for the queries that are not padding themselves
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int k0 = kbase + kt * 16 + lg * 4;      // this lane's four keys of the tile (a multiple of 4: one aligned 4-byte read)
                const unsigned km = k0 < nk ? *reinterpret_cast<const unsigned*>(kmask + (size_t)b * kmask_stride + k0) : 0xffffffffu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool pad = ((km >> (8 * r)) & 0xffu) == 0 && k0 + r < nk;
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) if (pad && q_valid[qt]) sc[qt][kt][r] = -1e30f;
                }
            }
        }
        // keys past nk (last stage) and, causal, keys after the query: score -1e30 -> probability exactly 0 once a real key set the maximum
        // (without a mask key 0 is in stage 0 for every query; with one a whole stage may be masked: p = 0 below while the maximum is -1e30)
        if (CAUSAL || kbase + EA_KSTAGE > nk) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int lim = CAUSAL ? min(nk - 1, q0 + wave * 32 + qt * 16 + lc + off) : nk - 1;   // last key this lane's query may see
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + kt * 16 + lg * 4 + r > lim) sc[qt][kt][r] = -1e30f;
            }
        }
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
                for (int r = 0; r < 4; ++r) {
                    // a masked score must give exactly 0 even while the running maximum is still the initial -1e30 (a query row
                    // beyond nq, clamped above, in a causal block): exp(-1e30 - (-1e30)) would be 1
                    const float p = sc[qt][kt][r] <= -1e30f ? 0.f : expf(sc[qt][kt][r] - m_new);
                    sc[qt][kt][r] = p; ps += p;
                }
            l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] *= alpha;
        }
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
        if (qrow < nq) {
            const int c0 = (piece * 4) ^ ((qq & 7) << 2);
            const float4 v4 = *reinterpret_cast<const float4*>(&tile[qq * 64 + c0]);
            TO* dst = out + ((size_t)(b * nq + qrow)) * inner + head * DH + piece * 4;
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

// rows b*t + (t - 1) of y [B*t][D] -> dense [B][D] (the last position of every image, for the single-position logits launch)
__global__ void gather_last_rows_kernel(const float* __restrict__ y, float* __restrict__ dst, int B, int t, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, c = i - b * D;
    dst[i] = y[((size_t)b * t + t - 1) * D + c];
}

// eos bookkeeping of the launch path rebuilt from tokens already generated (sliding-window continuation behind a persistent launch)
__global__ void rebuild_eos_state_kernel(const int64_t* __restrict__ tokens, int stride, int n, int rows, int eos, int bos, int* eos_seen, int* count) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int seen = (eos >= 0 && bos == eos) ? 1 : 0;
    for (int j = 0; j < n && !seen; ++j) seen = tokens[(size_t)r * stride + j] == eos;
    eos_seen[r] = seen;
    if (seen) atomicAdd(count, 1);
}

}  // namespace txo
