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
//
// gfx950 mapping: block = 4 waves, output tile = (16*MT rows) x 32 columns; the four waves split K in
// interleaved 64-byte chunks (each streams a distinct quarter of the weight rows straight from L2 into
// VGPRs -- weights are read once per block, no LDS round trip), partial tiles are summed through LDS and the
// epilogue runs on the reduced tile.  The normalised A rows (K = embed_dim) live in LDS, XOR-swizzled for
// conflict-free ds_read_b128 fragment reads.  The decode position t is read from device memory so that one
// captured launch sequence can be replayed for every step.
// Bound: latency / L2 weight streaming (M <= 64 per tile; the arithmetic is a few microseconds at most).
#pragma once
#include "common.h"

namespace txo {

enum { PRO_NONE = 0, PRO_EMBED = 1, PRO_LN2 = 2, PRO_LNF = 3 };
enum { EPI_QKV = 0, EPI_Q = 1, EPI_GLU_RES = 2, EPI_GEGLU = 3, EPI_BIAS_RES = 4, EPI_LOGITS = 5 };

template <typename T> struct DecGemmArgs {
    // problem
    int rows, N, K;                 // rows = batch, W is [N][K]
    const T* W;
    const float* bias;              // interleaved order for the paired epilogues
    // prologue
    const T* A;                     // PRO_NONE: [rows][K]
    const float* y;                 // PRO_LN2 / PRO_LNF: [rows][K] fp32 stream
    float* x_out;                   // PRO_EMBED / PRO_LN2: residual written here (block column 0 only)
    const float* gamma; const float* beta;
    const int64_t* tok; const float* tok_emb; const float* pos_emb;   // PRO_EMBED
    const int* t_ptr;               // decode position (device)
    // epilogue
    float* q_out;                   // [rows][inner] fp32
    T* k_cache; T* v_cache;         // [rows*heads][tmax][64]
    int inner, heads, tmax;
    const float* resid; float* y_out; int D;    // y_out [rows][D]
    T* h_out; int F;                // [rows][F]
    float* logits;                  // [rows][N]
};

template <int NVMAX>
__device__ inline void ln16(float4 (&v)[NVMAX], int nv, const float* gamma, const float* beta, int sub, float inv_d) {
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
    const float rstd = 1.0f / sqrtf(row16_sum(q) * inv_d + LN_EPS);
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + i * 64 + sub * 4);
        const float4 b = *reinterpret_cast<const float4*>(beta + i * 64 + sub * 4);
        v[i].x = v[i].x * rstd * g.x + b.x; v[i].y = v[i].y * rstd * g.y + b.y;
        v[i].z = v[i].z * rstd * g.z + b.z; v[i].w = v[i].w * rstd * g.w + b.w;
    }
}

// byte offset of element k of row r in the LDS A image (rows of K*sizeof(T) bytes, 16-byte pieces swizzled)
template <typename T>
__device__ inline int a_off(int r, int k, int row_bytes, int pmask) {
    const int piece = (k * (int)sizeof(T)) >> 4;
    return r * row_bytes + ((piece ^ (r & pmask)) << 4) + ((k * (int)sizeof(T)) & 15);
}

template <typename T, int MT, int PRO, int EPI, int NVMAX>
__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemmArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK, BM = 16 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * 32;
    const int K = a.K, rows = a.rows;
    const int row_bytes = K * (int)sizeof(T);
    const int pmask = min(16, row_bytes >> 4) - 1;
    int t = 0;
    if constexpr (PRO == PRO_EMBED || EPI == EPI_QKV) t = *a.t_ptr;

    // ------------------------------ prologue: normalised rows -> LDS ------------------------------
    if constexpr (PRO != PRO_NONE) {
        const int sub = tid & 15, nv = K >> 6;
        const float inv_d = 1.0f / K;
        for (int r = tid >> 4; r < BM; r += 16) {
            const int m = min(m0 + r, rows - 1);
            float4 v[NVMAX];
            if constexpr (PRO == PRO_EMBED) {
                const float* te = a.tok_emb + (size_t)a.tok[m] * K;
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
                    v[i] = *reinterpret_cast<const float4*>(a.y + (size_t)m * K + i * 64 + sub * 4);
            }
            if constexpr (PRO == PRO_LN2) ln16<NVMAX>(v, nv, a.gamma, a.beta, sub, inv_d);
            if constexpr (PRO == PRO_EMBED || PRO == PRO_LN2) {
                if (blockIdx.x == 0 && m0 + r < rows) {
#pragma unroll
                    for (int i = 0; i < NVMAX; ++i) if (i < nv)
                        *reinterpret_cast<float4*>(a.x_out + (size_t)m * K + i * 64 + sub * 4) = v[i];
                }
            }
            ln16<NVMAX>(v, nv, a.gamma, a.beta, sub, inv_d);
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv) {
                unsigned char* dst = smem + a_off<T>(r, i * 64 + sub * 4, row_bytes, pmask);
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(dst) = v[i];
                } else {
                    union { bf16 h[4]; uint2 u; } c;
                    c.h[0] = __float2bfloat16(v[i].x); c.h[1] = __float2bfloat16(v[i].y);
                    c.h[2] = __float2bfloat16(v[i].z); c.h[3] = __float2bfloat16(v[i].w);
                    *reinterpret_cast<uint2*>(dst) = c.u;
                }
            }
        }
        __syncthreads();
    }

    // ------------------------------ main: wave w owns k-chunks w, w+4, ... ------------------------------
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
    const int nch = K / KCH;
    const int wr0 = min(n0 + lr, a.N - 1), wr1 = min(n0 + 16 + lr, a.N - 1);
    const T* w0 = a.W + (size_t)wr0 * K + lg * PER16;
    const T* w1 = a.W + (size_t)wr1 * K + lg * PER16;
    for (int kc = wave; kc < nch; kc += 4) {
        const u32x4 fw0 = ld16(w0 + kc * KCH), fw1 = ld16(w1 + kc * KCH);
        u32x4 fa[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (PRO != PRO_NONE) {
                fa[i] = ld16(smem + a_off<T>(16 * i + lr, kc * KCH + lg * PER16, row_bytes, pmask));
            } else {
                const int m = min(m0 + 16 * i + lr, rows - 1);
                fa[i] = ld16(a.A + (size_t)m * K + kc * KCH + lg * PER16);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) { mma16<T>(acc[i][0], fa[i], fw0); mma16<T>(acc[i][1], fa[i], fw1); }
    }

    // ------------------------------ cross-wave K reduction through LDS ------------------------------
    __syncthreads();                                  // everyone is done reading the A image
    f32x4* red = reinterpret_cast<f32x4*>(smem);      // [wave][MT][2][64]
#pragma unroll
    for (int i = 0; i < MT; ++i) { red[((wave * MT + i) * 2 + 0) * 64 + lane] = acc[i][0];
                                   red[((wave * MT + i) * 2 + 1) * 64 + lane] = acc[i][1]; }
    __syncthreads();
    if (wave >= MT) return;
    const int mi = wave;
    f32x4 c0 = red[((0 * MT + mi) * 2 + 0) * 64 + lane], c1 = red[((0 * MT + mi) * 2 + 1) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { c0 += red[((w * MT + mi) * 2 + 0) * 64 + lane];
                                  c1 += red[((w * MT + mi) * 2 + 1) * 64 + lane]; }

    // ------------------------------ epilogue (C/D layout: col = lane&15, row = 4*(lane>>4)+reg) ------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * mi + lg * 4 + r;
        if (m >= rows) continue;
        const int na = n0 + lr, nb = n0 + 16 + lr;
        if constexpr (EPI == EPI_GLU_RES || EPI == EPI_GEGLU) {
            const int j = (n0 >> 1) + lr;             // 32 interleaved weight rows -> 16 outputs
            const float v = c0[r] + a.bias[na], g = c1[r] + a.bias[nb];
            if constexpr (EPI == EPI_GLU_RES) a.y_out[(size_t)m * a.D + j] = v * sigmoidf(g) + a.resid[(size_t)m * a.D + j];
            else a.h_out[(size_t)m * a.F + j] = Elem<T>::from_f32(v * gelu_erf(g));
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = h ? nb : na;
                const float v = h ? c1[r] : c0[r];
                if (n >= a.N) continue;
                if constexpr (EPI == EPI_QKV || EPI == EPI_Q) {
                    const int which = n / a.inner, f = n - which * a.inner;
                    if (which == 0) a.q_out[(size_t)m * a.inner + f] = v;
                    else {
                        T* cache = (which == 1) ? a.k_cache : a.v_cache;
                        cache[(((size_t)m * a.heads + (f >> 6)) * a.tmax + t) * DH + (f & 63)] = Elem<T>::from_f32(v);
                    }
                } else if constexpr (EPI == EPI_BIAS_RES) {
                    a.y_out[(size_t)m * a.D + n] = v + a.bias[n] + a.resid[(size_t)m * a.D + n];
                } else {   // EPI_LOGITS
                    a.logits[(size_t)m * a.N + n] = v + a.bias[n];
                }
            }
        }
    }
}

template <typename T, int MT>
inline size_t dec_gemm_lds_bytes(int K, bool has_pro) {
    const size_t red = (size_t)4 * MT * 2 * 64 * 16;
    const size_t img = has_pro ? (size_t)16 * MT * K * sizeof(T) : 0;
    return red > img ? red : img;
}

}  // namespace txo
