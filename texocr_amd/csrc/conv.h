// Hybrid CNN embedder (reference model/resnet.py:14-254 + model/encoder.py:31-72,162-169): the
// weight-standardised ResNetV2 [2,4,6] backbone that create_model(config) actually builds in front of the ViT.
//
//   StdConv2d  (resnet.py:38-66)  : conv with per-output-channel standardised weights and TF-"SAME" padding.
//                                   The standardisation depends only on the weights -> done once at load
//                                   (Engine::finalize); here a conv is an implicit GEMM through gemm_big_kernel:
//                                   A rows are gathered from the NHWC activation (k = (kh, kw, ic), ic contiguous).
//   GroupNormAct (resnet.py:14-35): GroupNorm(32, eps 1e-5, biased variance) + optional ReLU; the block's residual
//                                   add + ReLU (Bottleneck.forward :143-149) is fused into the last norm's apply pass.
//   MaxPool2d  (resnet.py:69-79)  : 3x3 / 2 with -inf SAME padding.
//
// Activations are NHWC of T (C innermost: a pixel's channels are one contiguous run, so 16-byte loads are
// channel vectors and 1x1 convs are plain row-major GEMMs).  GroupNorm statistics are fp32/fp64 and
// deterministic (fixed reduction order, no float atomics).  Bound: MFMA for the convs, HBM for the norms.
#pragma once
#include "common.h"
#include "gemm_big.h"

namespace txo {

// A loader for gemm_big_kernel: row m = (b, oh, ow), k = (kh, kw, ic)
template <typename T> struct LoadConv {
    const T* in; int H, W, C, stride, pt, pl;
    FastDiv d_ohw, d_ow, d_c, d_kw;
    __device__ inline u32x4 operator()(int m, int k) const {
        uint32_t b, r, oh, ow, kk, ic, kh, kw;
        d_ohw.divmod((uint32_t)m, b, r); d_ow.divmod(r, oh, ow);
        d_c.divmod((uint32_t)k, kk, ic); d_kw.divmod(kk, kh, kw);
        const int ih = (int)oh * stride - pt + (int)kh, iw = (int)ow * stride - pl + (int)kw;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) return u32x4{0u, 0u, 0u, 0u};
        return ld16(in + (((size_t)b * H + ih) * W + iw) * C + ic);
    }
};

// stem: 7x7 / 2 on the single-channel fp32 image; K = 49 padded to 64 with zero weights
template <typename T> struct LoadStem {
    const float* img; int H, W, pt, pl;
    FastDiv d_ohw, d_ow;
    __device__ inline u32x4 operator()(int m, int k) const {
        constexpr int PER16 = Elem<T>::PER16;
        uint32_t b, r, oh, ow;
        d_ohw.divmod((uint32_t)m, b, r); d_ow.divmod(r, oh, ow);
        float v[PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            const int kk = k + e, kh = kk / 7, kw = kk - kh * 7;
            const int ih = (int)oh * 2 - pt + kh, iw = (int)ow * 2 - pl + kw;
            const bool ok = kk < 49 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            v[e] = ok ? img[((size_t)b * H + ih) * W + iw] : 0.f;
        }
        if constexpr (sizeof(T) == 4) {
            return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        } else {
            union { bf16 h[8]; u32x4 u; } c;
#pragma unroll
            for (int e = 0; e < 8; ++e) c.h[e] = __float2bfloat16(v[e]);
            return c.u;
        }
    }
};

// x[b][1+p][n] = acc + bias[n] + pos[1 + pr*Gw + pc][n]  for a token grid hw = h*w inside a canvas grid Gw wide
// (HybridEmbedding.proj, encoder.py:61,71 + the position-id gather of VisionTransformer.forward :136-143)
struct EpiTokens {
    float* x; const float* bias; const float* pos; int D, hw, w, Gw;
    static constexpr bool PAIRED = false;
    static constexpr int ST = store8_insts<float>();   // 16-byte store instructions per fin() (gemm_pp.h)
    static constexpr int NCB = 8;
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        const int b = m / hw, p = m - b * hw, pr = p / w, pc = p - pr * w;
        float bb[8], pp[8];
        load8(bias + n, bb); load8(pos + (size_t)(1 + pr * Gw + pc) * D + n, pp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e] + pp[e];
        store8<float>(x + ((size_t)b * (hw + 1) + 1 + p) * D + n, v);
    }
    static constexpr bool HAS_ROW = true;
    __device__ inline void cols(int n, float (&cb)[32]) const { load8(bias + n, reinterpret_cast<float (&)[8]>(cb)); }
    __device__ inline void rowop(int m, int n, float (&r)[10]) const {
        const int b = m / hw, p = m - b * hw, pr = p / w, pc = p - pr * w;
        load8(pos + (size_t)(1 + pr * Gw + pc) * D + n, reinterpret_cast<float (&)[8]>(r));
    }
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&cb)[32], const float (&r)[10], bool valid) const {
        const int b = m / hw, p = m - b * hw;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += cb[e] + r[e];
        if (valid) store8<float>(x + ((size_t)b * (hw + 1) + 1 + p) * D + n, v);
    }
};

// ---- GroupNorm(32) ------------------------------------------------------------------------------------
// Stage 1: per (image, pixel chunk): partial sum / sum of squares of each of the 32 groups.
// 256*VEC is a multiple of C for every C in {64..1024}, so a thread always visits the same VEC channels and
// keeps private accumulators; they are combined through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                                         int chunk_px) {
    constexpr int VEC = Elem<T>::PER16;
    __shared__ float ps[256][2 * VEC];
    __shared__ float cs[1024][2];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, tid = threadIdx.x;
    const int px0 = chunk * chunk_px, px1 = min(HW, px0 + chunk_px);
    const int Q = C / VEC;                          // threads per pixel
    const int c0 = (tid % Q) * VEC, pstep = 256 / Q;
    float s[VEC], q[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int px = px0 + tid / Q; px < px1; px += pstep) {
        const u32x4 raw = ld16(x + ((size_t)b * HW + px) * C + c0);
        float f[VEC];
        if constexpr (VEC == 4) { f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y); f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w); }
        else {
            f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
            f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
            f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
            f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ps[tid][e] = s[e]; ps[tid][VEC + e] = q[e]; }
    __syncthreads();
    // per channel: sum over the 256/Q threads that own it (fixed order)
    for (int c = tid; c < C; c += 256) {
        const int owner = c / VEC, e = c - owner * VEC;
        float a = 0.f, d = 0.f;
        for (int i = 0; i < pstep; ++i) { a += ps[owner + Q * i][e]; d += ps[owner + Q * i][VEC + e]; }
        cs[c][0] = a; cs[c][1] = d;
    }
    __syncthreads();
    if (tid < 32) {
        const int cpg = C / 32;
        float a = 0.f, d = 0.f;
        for (int i = 0; i < cpg; ++i) { a += cs[tid * cpg + i][0]; d += cs[tid * cpg + i][1]; }
        float* dst = partial + (((size_t)b * nchunk + chunk) * 32 + tid) * 2;
        dst[0] = a; dst[1] = d;
    }
}

// Stage 2: per (image, group): mean and 1/sqrt(var + eps) from the chunk partials (fp64 combine).
__global__ void gn_finish_kernel(const float* __restrict__ partial, float* __restrict__ stats, int nchunk, double count) {
    const int b = blockIdx.x, g = threadIdx.x;
    if (g >= 32) return;
    double a = 0.0, d = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const float* p = partial + (((size_t)b * nchunk + c) * 32 + g) * 2;
        a += p[0]; d += p[1];
    }
    const double mean = a / count;
    const double var = fmax(d / count - mean * mean, 0.0);
    stats[((size_t)b * 32 + g) * 2 + 0] = (float)mean;
    stats[((size_t)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// Stage 3: y = (x - mean) * rstd * gamma + beta (+ residual) (ReLU)
template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* x, const T* res, T* y,   // y may alias x or res (same index per thread)
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW, int C, size_t nvec) {
    constexpr int VEC = Elem<T>::PER16;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const size_t e0 = i * VEC;
    const int c0 = (int)(e0 % C);
    const int b = (int)(e0 / ((size_t)HW * C));
    const int cpg = C / 32;
    const u32x4 raw = ld16(x + e0);
    u32x4 rr = {0u, 0u, 0u, 0u};
    if constexpr (RES) rr = ld16(res + e0);
    float f[VEC], r[VEC];
    if constexpr (VEC == 4) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y); f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
        r[0] = __uint_as_float(rr.x); r[1] = __uint_as_float(rr.y); r[2] = __uint_as_float(rr.z); r[3] = __uint_as_float(rr.w);
    } else {
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w}, v[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = __uint_as_float(w[k] << 16); f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
            r[2 * k] = __uint_as_float(v[k] << 16); r[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
        }
    }
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int c = c0 + e, g = c / cpg;
        const float mean = stats[((size_t)b * 32 + g) * 2], rstd = stats[((size_t)b * 32 + g) * 2 + 1];
        float v = (f[e] - mean) * rstd * gamma[c] + beta[c];
        if constexpr (RES) v += r[e];
        if constexpr (RELU) v = fmaxf(v, 0.f);
        o[e] = v;
    }
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(y + e0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        union { bf16 h[8]; u32x4 u; } c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c.h[e] = __float2bfloat16(o[e]);
        st16(y + e0, c.u);
    }
}

// MaxPool 3x3 / 2, SAME padding with -inf (resnet.py:69-79): NHWC, one thread per (pixel, channel vector)
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int OH,
                                                           int OW, int pt, int pl, size_t nvec) {
    constexpr int VEC = Elem<T>::PER16;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int cv = C / VEC;
    const int c0 = (int)(i % cv) * VEC;
    size_t p = i / cv;
    const int ow = (int)(p % OW); p /= OW;
    const int oh = (int)(p % OH);
    const int b = (int)(p / OH);
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = oh * 2 - pt + kh, iw = ow * 2 - pl + kw;
            if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
            const u32x4 raw = ld16(x + (((size_t)b * H + ih) * W + iw) * C + c0);
            if constexpr (VEC == 4) {
                m[0] = fmaxf(m[0], __uint_as_float(raw.x)); m[1] = fmaxf(m[1], __uint_as_float(raw.y));
                m[2] = fmaxf(m[2], __uint_as_float(raw.z)); m[3] = fmaxf(m[3], __uint_as_float(raw.w));
            } else {
                const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    m[2 * k] = fmaxf(m[2 * k], __uint_as_float(w4[k] << 16));
                    m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(w4[k] & 0xffff0000u));
                }
            }
        }
    T* dst = y + (((size_t)b * OH + oh) * OW + ow) * C + c0;
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(m[0], m[1], m[2], m[3]);
    else {
        union { bf16 h[8]; u32x4 u; } c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c.h[e] = __float2bfloat16(m[e]);
        st16(dst, c.u);
    }
}

}  // namespace txo
