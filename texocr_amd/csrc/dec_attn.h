// Decode-step attention: ONE query row per (image, head) against a cached K/V panel.
//
// Reference (KV-cached form of MultiHeadAttention.forward, model/attention.py:148-173):
//   cross : keys/values = per-layer projections of the raw encoder output, N = h*w+1 rows, non-causal.
//           The reference re-projects them from `enc` at every step (:124-126); here they are projected
//           once per generate() call and this kernel re-reads them every step -> the HBM-bound kernel of
//           the whole path.
//   self  : keys/values of positions 0..t (causal; with one query at the last position the triangular
//           fill of :158-163 masks nothing that is cached).
//   energy = q.k * 0.125 ; softmax (max-subtracted, fp32) ; out = sum_j p_j v_j ; heads merged
//   'b h n d -> b n (h d)' on the way out.
//
// Layout: K,V [images*heads][Lmax][64] of T (f32: 256-byte rows, bf16: 128-byte rows).
// gfx950 mapping: one 256-thread block per (image, head).  Every wave-instruction reads 1 KiB of contiguous
// K (or V) rows with 16 B per lane (LPR = 16 or 8 lanes per row), 8 such loads in flight per wave.  Scores go
// to LDS, the block does one max/sum pass, then the V sweep accumulates 16-byte column pieces per lane and
// reduces across lanes (shuffles) and waves (LDS).  Algorithmic bytes per launch = images*heads*2*L*64*sizeof(T).
// Bound: HBM (8 TB/s peak).
#pragma once
#include "common.h"

namespace txo {

template <typename T> struct DecAttnArgs {
    const float* q;          // [images][heads*64] fp32
    const T* K; const T* V;  // [images*heads][lmax][64]
    T* out;                  // [images][heads*64]
    int heads, lmax;
    int len;                 // number of keys, or -1: read *t_ptr + 1 (self attention at position t)
    const int* t_ptr;
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

template <typename T>
__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [len] scores, then [4][64] partial outputs + stats
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int LPR = DH / PER16;          // lanes per 64-element row: 16 (f32) / 8 (bf16)
    constexpr int KPI = 64 / LPR;            // keys per wave-instruction: 4 / 8
    constexpr int UNR = 8;
    const int bh = blockIdx.x, img = bh / a.heads, head = bh - img * a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPR, kq = lane / LPR;
    const int L = a.len >= 0 ? a.len : (*a.t_ptr + 1);
    const T* Kb = a.K + (size_t)bh * a.lmax * DH;
    const T* Vb = a.V + (size_t)bh * a.lmax * DH;
    __shared__ float part[4][DH];
    __shared__ float stat[8];

    // this lane's 16-byte piece of the query, pre-scaled (0.125 is exact)
    float qv[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) qv[e] = a.q[(size_t)img * a.heads * DH + head * DH + sub * PER16 + e] * ATTN_SCALE;

    // ---- phase 1: scores ----
    const int KPB = KPI * 4;                 // keys per block-iteration (all four waves)
    for (int base = 0; base < L; base += KPB * UNR) {
        u32x4 raw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int key = min(base + u * KPB + wave * KPI + kq, L - 1);
            raw[u] = ld16(Kb + (size_t)key * DH + sub * PER16);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float kf[PER16];
            unpack16<T, PER16>(raw[u], kf);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < PER16; ++e) d = fmaf(qv[e], kf[e], d);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            const int key = base + u * KPB + wave * KPI + kq;
            if (sub == 0 && key < L) sc[key] = d;
        }
    }
    __syncthreads();
    // ---- phase 2: softmax statistics ----
    float mx = -3.0e38f;
    for (int j = tid; j < L; j += 256) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if (lane == 0) stat[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
    float sum = 0.f;
    for (int j = tid; j < L; j += 256) { const float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) stat[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((stat[4] + stat[5]) + (stat[6] + stat[7]));

    // ---- phase 3: out = sum_j p_j v_j ----
    float acc[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) acc[e] = 0.f;
    for (int base = 0; base < L; base += KPB * UNR) {
        u32x4 raw[UNR];
        float pj[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int key = base + u * KPB + wave * KPI + kq;
            const int kc = min(key, L - 1);
            raw[u] = ld16(Vb + (size_t)kc * DH + sub * PER16);
            pj[u] = key < L ? sc[kc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float vf[PER16];
            unpack16<T, PER16>(raw[u], vf);
#pragma unroll
            for (int e = 0; e < PER16; ++e) acc[e] = fmaf(pj[u], vf[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < PER16; ++e) {
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < PER16; ++e) part[wave][sub * PER16 + e] = acc[e];
    }
    __syncthreads();
    if (tid < DH) {
        const float o = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) * inv;
        a.out[(size_t)img * a.heads * DH + head * DH + tid] = Elem<T>::from_f32(o);
    }
}

}  // namespace txo
