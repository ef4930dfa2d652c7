// Large-M GEMM for the encoder side of OCRModel.generate():  C[M,N] = A[M,K] * W[N,K]^T (+ epilogue).
//
// One kernel template, used for: patch embedding (reference model/encoder.py:25-28, A gathered
// straight from the NCHW image -- no im2col buffer), q/k/v projection (model/attention.py:124-127,
// epilogue scatters into head-major [B,heads,N,64]), gated output projection (:96-99,180, epilogue =
// bias + GLU + residual), GeGLU FFN-in (:15-17) and FFN-out (:63-67), and the one-off cross-attention
// K/V projection of the encoder output for every decoder layer (:125-126).
//
// gfx950 mapping: 256 threads = 4 waves as 2x2, block tile 128x128, wave tile 64x64 = 4x4 MFMA 16x16
// tiles (64 accumulator VGPRs).  K is consumed in 128-byte stages (32 f32 / 64 bf16 per row), staged
// global -> VGPR -> LDS with the next stage's global loads in flight behind the current stage's MFMAs;
// LDS rows are 128 B with an XOR piece swizzle so the ds_read_b128 fragment reads are conflict free.
// Weights are [N][K] (K contiguous) exactly as nn.Linear stores them, so A and W stage identically.
// Bound: MFMA (f32-in MFMA = 157 TF peak in fp32 parity mode, bf16 MFMA = 2.5 PF in bf16 mode).
#pragma once
#include "common.h"

namespace txo {

constexpr int GB_BM = 128, GB_BN = 128, GB_THREADS = 256, GB_STAGE_BYTES = 128;

// ---------------- A loaders: return 16 bytes (PER16 elements of T) of row m at element k ----------
template <typename T> struct LoadPlain {
    const T* A; int lda;
    __device__ inline u32x4 operator()(int m, int k) const { return ld16(A + (size_t)m * lda + k); }
};

// patch rows: m = (b, pr, pc), k = (c, py, px) -> img[b][c][pr*16+py][pc*16+px]  (fp32 NCHW image)
template <typename T> struct LoadPatch {
    const float* img; int C, H, W, hw, w;   // hw = patches per image, w = patches per row
    __device__ inline u32x4 operator()(int m, int k) const {
        int b = m / hw, p = m - b * hw, pr = p / w, pc = p - pr * w;
        int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
        const float* src = img + (((size_t)b * C + c) * H + pr * 16 + py) * W + pc * 16 + px;
        if constexpr (sizeof(T) == 4) {
            return ld16(src);
        } else {
            float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            union { bf16 h[8]; u32x4 v; } u;
            u.h[0] = __float2bfloat16(lo.x); u.h[1] = __float2bfloat16(lo.y); u.h[2] = __float2bfloat16(lo.z);
            u.h[3] = __float2bfloat16(lo.w); u.h[4] = __float2bfloat16(hi.x); u.h[5] = __float2bfloat16(hi.y);
            u.h[6] = __float2bfloat16(hi.z); u.h[7] = __float2bfloat16(hi.w);
            return u.v;
        }
    }
};

// ---------------- epilogues ----------------------------------------------------------------------------------
// The accumulator tile is staged through LDS and handed to the epilogue as 8 consecutive columns of one row
// (plain: v[8] at columns n..n+7; paired: value v[8] and gate g[8] of outputs j..j+7), so every global access is a
// 16- or 32-byte row segment instead of a 2-4 byte element of the MFMA register layout.
// nt: non-temporal stores.  The outputs of the encoder-side GEMMs are far larger than the caches and are next read by ANOTHER kernel; written
// with the default policy they are allocated in L2 and push the GEMM's own operand panels out of it (probes/pp_store_policy.hip, 150 784 rows,
// K = 768: 903-915 TFLOP/s with plain stores, 1094-1141 with non-temporal ones; K = 3072: no difference).  Wave-uniform flag.
// 16-byte store instructions ONE store8<TO>() issues: the epilogue functors derive their ST from it (gemm_pp.h counts a tile's stores in its
// relaxed seam wait, vmcnt(4 + stores): ST may understate -- the wait is then stricter than needed -- but must NEVER overstate)
template <typename TO> constexpr int store8_insts() { return sizeof(TO) == 4 ? 2 : 1; }
template <typename TO> __device__ inline void store8(TO* p, const float (&v)[8], int nt = 0) {
    if constexpr (sizeof(TO) == 4) {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        if (nt) { __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4)); }
        else { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
    } else {
        union { bf16 h[8]; u32x4 u; } c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c.h[e] = __float2bfloat16(v[e]);
        if (nt) __builtin_nontemporal_store(c.u, reinterpret_cast<u32x4*>(p)); else st16(p, c.u);
    }
}
__device__ inline void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T> struct EpiStore {           // out[m][n..] = acc (+ bias)
    T* out; int ldo; const float* bias; int nt = 0;
    static constexpr bool PAIRED = false;
    static constexpr int ST = store8_insts<T>();        // 16-byte store instructions per fin() (gemm_pp.h counts a tile's stores)
    static constexpr int NCB = 8;                       // entries of cb[] that cols() fills
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        if (bias) { float b[8]; load8(bias + n, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e]; }
        store8<T>(out + (size_t)m * ldo + n, v, nt);
    }
    // split form (gemm_pp.h): column operands once per lane, row operands batched ahead of the arithmetic, masked store
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int n, float (&cb)[32]) const {
        if (bias) load8(bias + n, reinterpret_cast<float (&)[8]>(cb));
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) cb[e] = 0.f; }
    }
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&cb)[32], const float (&)[10], bool valid) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += cb[e];
        if (valid) store8<T>(out + (size_t)m * ldo + n, v, nt);
    }
};
template <typename T> struct EpiHeads {           // scatter n = (which, head, d) into which-th [B,heads,Ntok,64]
    T* base; size_t which_stride; int inner, heads, ntok; int nt = 0;
    static constexpr bool PAIRED = false;
    static constexpr int ST = store8_insts<T>();
    static constexpr int NCB = 0;
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        const int which = n / inner, f = n - which * inner, head = f >> 6, d = f & 63;   // 8 columns never straddle a head
        const int b = m / ntok, t = m - b * ntok;
        store8<T>(base + which * which_stride + (((size_t)b * heads + head) * ntok + t) * DH + d, v, nt);
    }
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int, float (&)[32]) const {}
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&)[32], const float (&)[10], bool valid) const {
        if (valid) (*this)(m, n, v);
    }
};
// Residual of the two encoder-side epilogues below: resid[m][j] itself (stats == nullptr), or -- the encoder's stream, where
// x = LN(y) is never written (rows.h, MODE 3) -- rebuilt from the previous sub-layer's y and that row's {mean, rstd}:
// x = ln_apply(y, mean, rstd, gamma[j], beta[j]), the very expression the row kernel normalises with.  `resid` may then be the
// epilogue's own output buffer: every element is read and written by the same thread.
struct ResidLN {
    const float* resid; const float* stats; const float* gb; int D;     // gb: gamma[D] followed by beta[D]
    __device__ inline void load(int m, int j, float (&r)[10]) const {
        {   // the fp32 stream: read once per sub-layer (see common.h: ld16_once)
            const u32x4 a = ld16_once(resid + (size_t)m * D + j), b = ld16_once(resid + (size_t)m * D + j + 4);
            r[0] = __uint_as_float(a.x); r[1] = __uint_as_float(a.y); r[2] = __uint_as_float(a.z); r[3] = __uint_as_float(a.w);
            r[4] = __uint_as_float(b.x); r[5] = __uint_as_float(b.y); r[6] = __uint_as_float(b.z); r[7] = __uint_as_float(b.w);
        }
        if (stats) { const float2 st = *reinterpret_cast<const float2*>(stats + (size_t)m * 2); r[8] = st.x; r[9] = st.y; }
    }
    // cg: gamma[j..j+7], beta[j..j+7] (loaded once per lane by cols())
    __device__ inline void cols(int j, float (&cg)[16]) const {
        if (stats) { load8(gb + j, reinterpret_cast<float (&)[8]>(cg)); load8(gb + D + j, reinterpret_cast<float (&)[8]>(cg[8])); }
    }
    __device__ inline float value(const float (&r)[10], const float (&cg)[16], int e) const {
        return stats ? ln_apply(r[e], r[8], r[9], cg[e], cg[8 + e]) : r[e];
    }
};
template <bool FAST = false>                      // FAST: perf mode (bf16 operands), see common.h
struct EpiGluRes {                                // y[m][j..] = (v+bv) * sigmoid(g+bg) + resid[m][j..]   (fp32 stream)
    float* y; ResidLN res; const float* bias;     // bias is in the interleaved order
    int nt = 0;
    static constexpr bool PAIRED = true;
    static constexpr int ST = store8_insts<float>();    // one store8<float> per fin()
    static constexpr int NCB = 32;
    __device__ inline void operator()(int m, int j, int nv, int ng, float (&v)[8], const float (&g)[8]) const {
        float bv[8], bg[8], r[10], cg[16];
        load8(bias + nv, bv); load8(bias + ng, bg); res.load(m, j, r); res.cols(j, cg);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[e]) * sigmoid_sel<FAST>(g[e] + bg[e]) + res.value(r, cg, e);
        store8<float>(y + (size_t)m * res.D + j, v, nt);
    }
    static constexpr bool HAS_ROW = true;
    __device__ inline void cols(int nv, int ng, float (&cb)[32]) const {
        load8(bias + nv, reinterpret_cast<float (&)[8]>(cb)); load8(bias + ng, reinterpret_cast<float (&)[8]>(cb[8]));
        res.cols((nv >> 5) * 16 + (nv & 15), reinterpret_cast<float (&)[16]>(cb[16]));
    }
    __device__ inline void rowop(int m, int j, float (&r)[10]) const { res.load(m, j, r); }
    __device__ inline void fin(int m, int j, float (&v)[8], const float (&g)[8], const float (&cb)[32], const float (&r)[10], bool valid) const {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (v[e] + cb[e]) * sigmoid_sel<FAST>(g[e] + cb[8 + e]) + res.value(r, reinterpret_cast<const float (&)[16]>(cb[16]), e);
        if (valid) store8<float>(y + (size_t)m * res.D + j, v, nt);
    }
};
template <typename T> struct EpiGeglu {           // h[m][j..] = (v+bv) * gelu(g+bg)
    T* h; const float* bias; int F; int nt = 0;
    static constexpr bool PAIRED = true;
    static constexpr int ST = store8_insts<T>();
    static constexpr int NCB = 16;
    __device__ inline void operator()(int m, int j, int nv, int ng, float (&v)[8], const float (&g)[8]) const {
        float bv[8], bg[8];
        load8(bias + nv, bv); load8(bias + ng, bg);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[e]) * gelu_sel<sizeof(T) == 2>(g[e] + bg[e]);
        store8<T>(h + (size_t)m * F + j, v, nt);
    }
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int nv, int ng, float (&cb)[32]) const {
        load8(bias + nv, reinterpret_cast<float (&)[8]>(cb)); load8(bias + ng, reinterpret_cast<float (&)[8]>(cb[8]));
    }
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int j, float (&v)[8], const float (&g)[8], const float (&cb)[32], const float (&)[10], bool valid) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + cb[e]) * gelu_sel<sizeof(T) == 2>(g[e] + cb[8 + e]);
        if (valid) store8<T>(h + (size_t)m * F + j, v, nt);
    }
};
struct EpiBiasRes {                               // y[m][n..] = acc + bias + resid
    float* y; ResidLN res; const float* bias; int nt = 0;
    static constexpr bool PAIRED = false;
    static constexpr int ST = store8_insts<float>();    // one store8<float> per fin()
    static constexpr int NCB = 24;
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        float b[8], r[10], cg[16];
        load8(bias + n, b); res.load(m, n, r); res.cols(n, cg);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b[e] + res.value(r, cg, e);
        store8<float>(y + (size_t)m * res.D + n, v, nt);
    }
    static constexpr bool HAS_ROW = true;
    __device__ inline void cols(int n, float (&cb)[32]) const {
        load8(bias + n, reinterpret_cast<float (&)[8]>(cb)); res.cols(n, reinterpret_cast<float (&)[16]>(cb[8]));
    }
    __device__ inline void rowop(int m, int n, float (&r)[10]) const { res.load(m, n, r); }
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&cb)[32], const float (&r)[10], bool valid) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += cb[e] + res.value(r, reinterpret_cast<const float (&)[16]>(cb[8]), e);
        if (valid) store8<float>(y + (size_t)m * res.D + n, v, nt);
    }
};
struct EpiPatch {                                 // x[b][1+p][n..] = acc + bias + pos[1 + pr*G + pc][n..]
    float* x; const float* bias; const float* pos; int D, hw, w, G;
    static constexpr bool PAIRED = false;
    __device__ inline void operator()(int m, int n, float (&v)[8]) const {
        const int b = m / hw, p = m - b * hw, pr = p / w, pc = p - pr * w;
        float bb[8], pp[8];
        load8(bias + n, bb); load8(pos + (size_t)(1 + pr * G + pc) * D + n, pp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e] + pp[e];
        store8<float>(x + ((size_t)b * (hw + 1) + 1 + p) * D + n, v);
    }
};

template <typename T, class ALoad, class Epi>
__global__ __launch_bounds__(GB_THREADS) void gemm_big_kernel(ALoad aload, const T* __restrict__ W, int M, int N,
                                                              int K, int tiles_n, int n_tiles, Epi epi) {
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int STAGE_K = GB_STAGE_BYTES / sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][GB_BM * 128];   // [buf][A|W][row*128]

    // XCD-aware tile order: blocks that share an XCD (blockIdx % 8) walk consecutive tiles of one
    // A row panel, so the panel is fetched into that XCD's L2 once (speed only; bijective remap).
    int bid = blockIdx.x;
    {
        int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * GB_BM, n0 = tile_n * GB_BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 15, lg = lane >> 4;

    // staging map: thread loads rows srow + 32*i (i<4), 16-byte piece spiece of the 128-byte stage row
    const int srow = tid >> 3, spiece = tid & 7;
    u32x4 ra[4], rw[4];
    const u32x4 zero = {0u, 0u, 0u, 0u};

    auto load_stage = [&](int kt) {
        const int k = kt * STAGE_K + spiece * PER16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + srow + 32 * i, n = n0 + srow + 32 * i;
            ra[i] = (m < M) ? aload(m, k) : zero;
            rw[i] = (n < N) ? ld16(W + (size_t)n * K + k) : zero;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            st16(&lds[buf][0][swz128(row, spiece)], ra[i]);
            st16(&lds[buf][1][swz128(row, spiece)], rw[i]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / STAGE_K;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_stage(kt + 1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {          // two 64-byte k-chunks per stage
            u32x4 fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = ld16(&lds[buf][0][swz128(wm + 16 * i + lr, kc * 4 + lg)]);
                fw[i] = ld16(&lds[buf][1][swz128(wn + 16 * i + lr, kc * 4 + lg)]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], fa[i], fw[j]);
        }
        if (kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS tile [128][128] f32 (reuses the staging buffers: exactly 64 KB) -> row segments.
    // C/D layout: col = lane&15, row = (lane>>4)*4 + reg.  Column XOR by 16 on odd 4-row groups keeps the 4-byte LDS
    // writes at the inherent 2 lanes per bank; 8-column groups stay contiguous for the row-wise reads.
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]);
    auto tidx = [](int row, int col) { return row * GB_BN + (col ^ (((row >> 2) & 1) << 4)); };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[tidx(wm + 16 * i + lg * 4 + r, wn + 16 * j + lr)] = acc[i][j][r];
    __syncthreads();
    if constexpr (Epi::PAIRED) {
        // columns interleaved in groups of 16: [16 value | 16 gate] per 32 weight rows -> 8 value groups of 8 per row
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + GB_THREADS * q, row = idx >> 3, vg = idx & 7;       // vg: value group within the row
            const int cv = (vg >> 1) * 32 + (vg & 1) * 8, m = m0 + row;
            const int nv = n0 + cv, ng = nv + 16;
            if (m < M && ng + 7 < N) {
                float v[8], g[8];
                load8(&tile[tidx(row, cv)], v); load8(&tile[tidx(row, cv + 16)], g);
                epi(m, (nv >> 5) * 16 + (nv & 15), nv, ng, v, g);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + GB_THREADS * q, row = idx >> 4, cg = idx & 15;
            const int m = m0 + row, n = n0 + cg * 8;
            if (m < M && n + 7 < N) {
                float v[8];
                load8(&tile[tidx(row, cg * 8)], v);
                epi(m, n, v);
            }
        }
    }
}

template <typename T, class ALoad, class Epi>
inline void launch_gemm_big(hipStream_t s, ALoad aload, const T* W, int M, int N, int K, Epi epi) {
    const int tiles_m = (M + GB_BM - 1) / GB_BM, tiles_n = (N + GB_BN - 1) / GB_BN;
    const int n_tiles = tiles_m * tiles_n;
    // the epilogue hands out 8-column row segments: every N on this path is a multiple of 8 (64-wide heads, 16-interleave)
    hipLaunchKernelGGL((gemm_big_kernel<T, ALoad, Epi>), dim3(n_tiles), dim3(GB_THREADS), 0, s, aload, W, M, N, K,
                       tiles_n, n_tiles, epi);
}

}  // namespace txo
