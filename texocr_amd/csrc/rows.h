// Row-wise kernels of the encoder stack: the shared-LayerNorm "sandwich" of AttentionLayers.forward
// (reference model/attention.py:242-259) and the CLS/pos-embed row (model/encoder.py:133-143).
//
//   residual = x; z = LN(x); out = block(z); x = out + residual; if not last: x = LN(x)
//
// so between two blocks the stream is normalised TWICE with the SAME (gamma, beta): x_next = LN(y) is the
// next residual, z = LN(x_next) the next block input.  One wave per row (D = 256 -> one 16-byte load per
// lane), row statistics by wave shuffles, two-pass variance in registers.  HBM-bound (reads D*4 B, writes
// D*4 + D*sizeof(T) per row).
#pragma once
#include "common.h"

namespace txo {

// LN of a row held as NV float4 per lane (row length D = NV * 256); returns normalised values in place.
// mean_out / rstd_out: the row statistics, for consumers that rebuild the normalised row themselves (ln_apply)
template <int NV>
__device__ inline void ln_row(float4 (&v)[NV], const float4 (&g)[NV], const float4 (&b)[NV], float inv_d, float* mean_out = nullptr,
                              float* rstd_out = nullptr) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float tx = v[i].x - mean, ty = v[i].y - mean, tz = v[i].z - mean, tw = v[i].w - mean;
        q += (tx * tx + ty * ty) + (tz * tz + tw * tw);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + LN_EPS);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x = ln_apply(v[i].x, mean, rstd, g[i].x, b[i].x); v[i].y = ln_apply(v[i].y, mean, rstd, g[i].y, b[i].y);
        v[i].z = ln_apply(v[i].z, mean, rstd, g[i].z, b[i].z); v[i].w = ln_apply(v[i].w, mean, rstd, g[i].w, b[i].w);
    }
    if (mean_out) { *mean_out = mean; *rstd_out = rstd; }
}

template <typename T>
__device__ inline void store4(T* p, const float4& v);
template <> __device__ inline void store4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ inline void store4<bf16>(bf16* p, const float4& v) {
    union { bf16 h[4]; uint2 u; } t;
    t.h[0] = __float2bfloat16(v.x); t.h[1] = __float2bfloat16(v.y);
    t.h[2] = __float2bfloat16(v.z); t.h[3] = __float2bfloat16(v.w);
    *reinterpret_cast<uint2*>(p) = t.u;
}

// MODE 0: x given      -> z = LN(x)                       (first sub-layer of a stack)
// MODE 1: y given      -> x = LN(y) (written), z = LN(x)  (between sub-layers)
// MODE 2: y given      -> z = LN_final(y)                 (after the stack; separate gamma/beta)
// MODE 3: y given      -> stats[row] = {mean, rstd} of LN(y) (written where MODE 1 writes x), z = LN(LN(y)): the encoder's form --
//         x itself is never materialised, the next GEMM epilogue rebuilds its residual from y and the statistics (ln_apply):
//         a third of this kernel's bytes (it is HBM-bound) and an M x D fp32 write less per sub-layer
template <typename T, int NV, int MODE>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ in, float* __restrict__ x_out,
                                                      T* __restrict__ z_out, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int rows, int rev) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // rev: rows in descending order -- the rows a GEMM wrote LAST (highest: its tiles walk the row panels upwards) are read first, while
    // they are still in the Infinity Cache, and the first rows of z, which the next GEMM reads first, are written last
    if (rev) row = rows - 1 - row;
    constexpr int D = NV * 256;
    const float inv_d = 1.0f / D;
    float4 v[NV], g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        v[i] = __builtin_bit_cast(float4, MODE == 3 ? ld16_once(in + (size_t)row * D + c) : ld16(in + (size_t)row * D + c));
        g[i] = *reinterpret_cast<const float4*>(gamma + c);
        b[i] = *reinterpret_cast<const float4*>(beta + c);
    }
    if constexpr (MODE == 3) {
        float mean, rstd;
        ln_row<NV>(v, g, b, inv_d, &mean, &rstd);
        if (lane == 0) *reinterpret_cast<float2*>(x_out + (size_t)row * 2) = make_float2(mean, rstd);
        ln_row<NV>(v, g, b, inv_d);
    } else {
        ln_row<NV>(v, g, b, inv_d);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(x_out + (size_t)row * D + i * 256 + lane * 4) = v[i];
            ln_row<NV>(v, g, b, inv_d);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) store4<T>(z_out + (size_t)row * D + i * 256 + lane * 4, v[i]);
}

// Generic-D fallback (D not a multiple of 256, e.g. the 64-wide test model): one wave per row, strided.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void ln_rows_generic_kernel(const float* __restrict__ in, float* __restrict__ x_out,
                                                              T* __restrict__ z_out, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float inv_d = 1.0f / D;
    const float* src = in + (size_t)row * D;
    if constexpr (MODE == 3) {                        // statistics of LN(y) out, x = LN(y) rebuilt per use, z = LN(x)
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += src[c];
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + LN_EPS);
        if (lane == 0) *reinterpret_cast<float2*>(x_out + (size_t)row * 2) = make_float2(mean, rstd);
        float s2 = 0.f;
        for (int c = lane; c < D; c += 64) s2 += ln_apply(src[c], mean, rstd, gamma[c], beta[c]);
        const float mean2 = wave_sum(s2) * inv_d;
        float q2 = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = ln_apply(src[c], mean, rstd, gamma[c], beta[c]) - mean2; q2 += d * d; }
        const float rstd2 = 1.0f / sqrtf(wave_sum(q2) * inv_d + LN_EPS);
        for (int c = lane; c < D; c += 64)
            z_out[(size_t)row * D + c] = Elem<T>::from_f32(ln_apply(ln_apply(src[c], mean, rstd, gamma[c], beta[c]), mean2, rstd2, gamma[c], beta[c]));
        return;
    }
    for (int pass = 0; pass < (MODE == 1 ? 2 : 1); ++pass) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += src[c];
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + LN_EPS);
        const bool to_x = (MODE == 1 && pass == 0);
        for (int c = lane; c < D; c += 64) {
            const float o = ln_apply(src[c], mean, rstd, gamma[c], beta[c]);
            if (to_x) x_out[(size_t)row * D + c] = o;
            else z_out[(size_t)row * D + c] = Elem<T>::from_f32(o);
        }
        // pass 2 re-reads exactly the elements this lane just wrote (same-thread program order)
        if (to_x) src = x_out + (size_t)row * D;
    }
}

// x[b][0][:] = cls + pos_embed[0]   (encoder.py:133-134,143; position id of the CLS token is 0)
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                int B, int ntok, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, c = i - b * D;
    x[(size_t)b * ntok * D + c] = cls[c] + pos[c];
}

}  // namespace txo
