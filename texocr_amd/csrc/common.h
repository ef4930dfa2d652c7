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

// ---- cross-lane exchange without the LDS crossbar -----------------------------------------------------------------
// hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-pipe operation, ~100 cycles of latency); the reductions of this
// path are chains of 4-6 DEPENDENT exchanges inside latency-bound kernels (LayerNorm prologues, the decode attention's
// score butterflies and accumulator sums, arg-max), so they are built from VALU-only exchanges instead:
//   partner at lane ^ 1, ^ 2        DPP quad_perm
//   partner 7 - i inside 8 lanes    DPP row_half_mirror        (after the two quad steps every lane of a quad holds the
//   partner 15 - i inside 16 lanes  DPP row_mirror              same value, so WHICH lane of the partner quad is read does
//   partner row (16 lanes) / half   v_permlane16_swap / v_permlane32_swap     not matter: same sums as a xor butterfly)
//   partner at lane ^ 8             DPP row_ror:8 (a rotation by 8 inside a row of 16 IS xor 8)
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
#ifdef TXO_SHFL_REDUCE   // diagnostic build: the same exchanges through ds_bpermute
template <int CTRL> __device__ inline int dpp_mov(int v) {
    const int l = threadIdx.x & 63;
    const int src = CTRL == DPP_XOR1 ? l ^ 1 : CTRL == DPP_XOR2 ? l ^ 2 : CTRL == DPP_HALF_MIRROR ? (l & ~7) | (7 - (l & 7))
                  : CTRL == DPP_MIRROR ? (l & ~15) | (15 - (l & 15)) : l ^ 8;
    return __shfl(v, src, 64);
}
template <int CTRL> __device__ inline float dpp_mov(float v) { return __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(int, v))); }
#else
template <int CTRL> __device__ inline float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ inline int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
#endif
// the other 16-lane row of a row pair (lane ^ 16) / the other 32-lane half (lane ^ 32): a = the even rows' (lower half's) value,
// b = the odd rows' (upper half's), in every lane of the pair.  The two results are copied into scalars BEFORE the bit cast:
// this clang's __builtin_bit_cast(float, r[1]) on an element of the returned vector reads element 0 (the index is dropped),
// which silently turns every reduction below into v + v (probes/reduce_check.hip, probes/swap_raw.hip).
__device__ inline void swap16(float v, float& a, float& b) {
#ifdef TXO_SHFL_REDUCE
    { const float o = __shfl_xor(v, 16, 64); const bool odd = (threadIdx.x >> 4) & 1; a = odd ? o : v; b = odd ? v : o; return; }
#endif
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
__device__ inline void swap32(float v, float& a, float& b) {
#ifdef TXO_SHFL_REDUCE
    { const float o = __shfl_xor(v, 32, 64); const bool hi = (threadIdx.x >> 5) & 1; a = hi ? o : v; b = hi ? v : o; return; }
#endif
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
// value of the lane at (lane ^ 16) / (lane ^ 32): after the swap one of the two results holds the partner row / half for the
// even rows (lower half) and the other for the odd rows (upper half)
__device__ inline float xor16(float v) { float a, b; swap16(v, a, b); return ((threadIdx.x >> 4) & 1) ? a : b; }
__device__ inline float xor32(float v) { float a, b; swap32(v, a, b); return ((threadIdx.x >> 5) & 1) ? a : b; }
__device__ inline int xor16(int v) { return __builtin_bit_cast(int, xor16(__builtin_bit_cast(float, v))); }
__device__ inline int xor32(int v) { return __builtin_bit_cast(int, xor32(__builtin_bit_cast(float, v))); }

// ---- wave reductions (every lane of the reduced group ends up with the result) ------------------------------------
// The library is built with -ffp-contract=on (build.py): a multiply and an add fuse only inside one source expression,
// so these adds are never fused with the multiply that produced their operand, and the same tile function gives the same
// bits in every kernel it is inlined into (the per-stage kernels and the persistent decode kernel).
// reduce over the 4 lanes of a quad / the 2 lanes (lane, lane ^ 8)
__device__ inline float quad_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v);
    return v;
}
__device__ inline float xor8_sum(float v) {
    return v + dpp_mov<DPP_ROR8>(v);
}
// reduce over the 8 lanes that share (lane >> 3)
__device__ inline float row8_sum(float v) {
    v = quad_sum(v); v += dpp_mov<DPP_HALF_MIRROR>(v);
    return v;
}
// reduce over the 16 lanes that share (lane >> 4)
__device__ inline float row16_sum(float v) {
    v = row8_sum(v); v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ inline float row16_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v)); v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
// reduce over the 4 lane groups (lanes l, l^16, l^32, l^48)
__device__ inline float grp4_sum(float v) {
    float a, b;
    swap16(v, a, b); v = a + b;
    swap32(v, a, b); return a + b;
}
__device__ inline float grp4_max(float v) {
    float a, b;
    swap16(v, a, b); v = fmaxf(a, b);
    swap32(v, a, b); return fmaxf(a, b);
}
__device__ inline float wave_sum(float v) { return grp4_sum(row16_sum(v)); }
__device__ inline float wave_max(float v) { return grp4_max(row16_max(v)); }
__device__ inline int wave_sum(int v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
    v += xor16(v); v += xor32(v);
    return v;
}
// arg-max over the wave: largest value, lowest index among equals (torch.argmax); the combine is commutative and
// associative, so the mirrored partners above serve
__device__ inline void wave_argmax(float& best, int& bi) {
    auto take = [&](float ov, int oi) { if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; } };
    take(dpp_mov<DPP_XOR1>(best), dpp_mov<DPP_XOR1>(bi));
    take(dpp_mov<DPP_XOR2>(best), dpp_mov<DPP_XOR2>(bi));
    take(dpp_mov<DPP_HALF_MIRROR>(best), dpp_mov<DPP_HALF_MIRROR>(bi));
    take(dpp_mov<DPP_MIRROR>(best), dpp_mov<DPP_MIRROR>(bi));
    take(xor16(best), xor16(bi));
    take(xor32(best), xor32(bi));
}

// 1 / sqrt(x): IEEE sqrt and division in the fp32 parity mode; perf mode (bf16 operands) takes v_rsq_f32 (1 ulp) -- the decode
// step evaluates ~25 LayerNorms per position on its critical path, and the exact form is a chain of ~25 dependent instructions
template <bool FAST> __device__ inline float rsqrt_sel(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rsqf(x); else return 1.0f / sqrtf(x);
}

// exp(x): libm in the fp32 parity mode; perf mode (bf16 operands) takes v_exp_f32 on x * log2(e) (the decode attention evaluates one
// per key slot on its critical path; exp(-huge) is still exactly 0, which the masking relies on)
template <bool FAST> __device__ inline float exp_sel(float x) {
    if constexpr (FAST) return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); else return expf(x);
}

// one LayerNorm element from its row statistics: the row kernels (rows.h) and the GEMM epilogues that rebuild the residual
// x = LN(y) on the fly (gemm_big.h: EpiGluRes / EpiBiasRes with stats) share this form, so both give the same bits
__device__ inline float ln_apply(float y, float mean, float rstd, float g, float b) { const float t = y - mean; return t * rstd * g + b; }

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
// experiment build (-DTXO_EXP_MALL=1): encoder-side streams that are dead after one use (the fp32 stream y read by the LayerNorm pass and by
// the residual epilogues, q/k/v read by the attention) are requested non-temporally, so that the 231 MB bf16 operand the NEXT GEMM reads (z, the
// attention output) survives in the 256 MB Infinity Cache
#ifndef TXO_EXP_MALL
#define TXO_EXP_MALL 0
#endif
__device__ inline u32x4 ld16_once(const void* p) { if constexpr (TXO_EXP_MALL) return ld16_stream(p); else return ld16(p); }

// 16 bytes from an LDS BYTE ADDRESS (not a generic pointer: a ds_read_b128 for sure, never a flat load)
__device__ inline float4 lds_ld_f4(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(3))) float4*>((uintptr_t)addr);
#else
    (void)addr; return float4{};
#endif
}

}  // namespace txo
