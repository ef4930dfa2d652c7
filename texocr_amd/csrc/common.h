// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16, LDS swizzles).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace txo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr float LN_EPS = 1e-5f;         // nn.LayerNorm default (reference attention.py:200)
constexpr float ATTN_SCALE = 0.125f;    // dim_head ** -0.5 with dim_head = 64 (attention.py:76,80)
constexpr int DH = 64;

// ---- element types ------------------------------------------------------------------
// A "k-chunk" is 64 bytes of one operand row: 16 f32 or 32 bf16.  One 16-byte piece of it per
// lane group (lane>>4) is exactly one MFMA operand fragment in both element types:
//   f32 : 4 floats  -> four v_mfma_f32_16x16x4_f32   (k = 4*group + e, e = 0..3; A and B use the
//                      same permuted k order, so the sum over k is unchanged)
//   bf16: 8 bf16    -> one  v_mfma_f32_16x16x32_bf16 (k = 8*group + j)
// C/D layout (both): col = lane & 15, row = (lane >> 4) * 4 + reg.
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;     // elements per 16 bytes
    static constexpr int KCHUNK = 16;   // elements per 64-byte k-chunk
    __device__ static inline float from_f32(float v) { return v; }
    __device__ static inline float to_f32(float v) { return v; }
};
template <> struct Elem<__hip_bfloat16> {
    static constexpr int PER16 = 8;
    static constexpr int KCHUNK = 32;
    __device__ static inline __hip_bfloat16 from_f32(float v) { return __float2bfloat16(v); }
    __device__ static inline float to_f32(__hip_bfloat16 v) { return __bfloat162float(v); }
};
typedef __hip_bfloat16 bf16;

template <typename T>
__device__ inline void mma16(f32x4& acc, const u32x4& a, const u32x4& b);

template <>
__device__ inline void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <>
__device__ inline void mma16<bf16>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
}

// ---- wave reductions ------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 16 lanes that share (lane >> 4)
__device__ inline float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// reduce over the 4 lane groups (lanes l, l^16, l^32, l^48)
__device__ inline float grp4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ inline float grp4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// exact-erf GELU (F.gelu default, reference attention.py:17) and sigmoid (nn.GLU, :98)
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// perf mode (bf16 operands) versions for the encoder-side GEMM epilogues, where the exact forms are a quarter of the FFN-in
// kernel (erff ~30 VALU instructions per output on top of the MFMAs, no matrix work to hide behind):
//   erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below one bf16 rounding of the result) on v_rcp_f32 / v_exp_f32.
// The fp32 parity mode keeps erff / expf / the IEEE division.
__device__ inline float gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f, a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);                       // erf(|z|)
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}
__device__ inline float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f)); }
template <bool FAST> __device__ inline float gelu_sel(float x) { if constexpr (FAST) return gelu_fast(x); else return gelu_erf(x); }
template <bool FAST> __device__ inline float sigmoid_sel(float x) { if constexpr (FAST) return sigmoid_fast(x); else return sigmoidf(x); }

// 128-byte LDS rows of 8 x 16-byte pieces, piece index XOR-swizzled with the row so that a
// ds_read_b128 fragment read (16 rows x one piece per lane group) is bank-conflict free.
__device__ inline int swz128(int row, int piece) { return row * 128 + ((piece ^ (row & 7)) << 4); }

// Division by a launch-time constant without an integer divide (the conv loaders split a GEMM row / k index
// into (image, oh, ow) / (kh, kw, ic) for every 16-byte load): q = (umulhi(n, m) + n) >> l, exact for n < 2^31.
struct FastDiv {
    uint32_t d, m, l;
    FastDiv() : d(1), m(1), l(0) {}
    explicit FastDiv(uint32_t div) : d(div) {
        l = 0;
        while ((1u << l) < div) ++l;
        m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - div)) / div + 1);
    }
    __host__ __device__ inline uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)(((uint64_t)__umulhi(n, m) + n) >> l);
#else
        return (uint32_t)(((((uint64_t)n * m) >> 32) + n) >> l);
#endif
    }
    __host__ __device__ inline void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};

__device__ inline u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
// Streaming (non-temporal) 16-byte load for data that is read once per launch and is far larger than the caches -- the
// decode-step K/V panels (~440 MB per step).  Measured with in-kernel stamps: a decode GEMM block waits 1.6-2.0 us for
// 48 KB of operands that come cold from HBM but 0.43 us when they are still cached; with plain loads the K/V stream
// evicts the 18 MB of decoder weights from the 256 MB Infinity Cache on every step.
__device__ inline u32x4 ld16_stream(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ inline void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

}  // namespace txo
