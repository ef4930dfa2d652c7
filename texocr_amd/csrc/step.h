// Greedy token selection + loop bookkeeping of AutoRegressiveDecoder.generate (reference
// model/decoder.py:103-116), kept entirely on the device so that a step never syncs with the host.
//
//   next = argmax(logits[:, -1, :])   -- "greedy": the reference samples (top-k -> softmax(/temp) ->
//                                        multinomial, :104-108); argmax is its temp->0 limit
//   output = cat(output, next)        -- tokens_out[row][t]
//   break if (output == eos).any(dim=1).all()   -- a GLOBAL condition over rows (:115-116): recorded as
//                                        done_flag[t]; the host reads the flags every few steps and trims.
//
// One wave per row.  The last row-block to arrive (one packed atomic: low 16 bits = arrivals, high 16 =
// rows that saw eos for the first time in this step) advances the device-side step counter.
#pragma once
#include "common.h"

namespace txo {

struct StepState {
    int t;              // position being decoded (0 = BOS)
    int rows_with_eos;  // rows that contain eos so far
    unsigned arrive;    // packed arrival word, 0 between launches
    int pad;
};

struct StepArgs {
    const float* logits; int V; int rows;
    int64_t* cur_tok;           // [rows] token fed to the next step
    int64_t* tokens_out;        // [rows][out_stride] or null
    int out_stride;
    float* logits_out;          // [rows][out_stride][V] or null
    StepState* st; int* eos_seen; int* done_flag; int eos;   // eos < 0: no eos check
};

__global__ __launch_bounds__(64) void argmax_step_kernel(StepArgs a) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int t = a.st->t;
    const float* lg = a.logits + (size_t)row * a.V;
    float best = -3.4e38f; int bi = 0x7fffffff;
    float* lo = a.logits_out ? a.logits_out + ((size_t)row * a.out_stride + t) * a.V : nullptr;
    if ((a.V & 3) == 0) {                      // rows are 16-byte aligned: four float4 per lane in flight
        const int n4 = a.V >> 2;
        for (int base = 0; base < n4; base += 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j4 = base + u * 64 + lane;
                v[u] = j4 < n4 ? reinterpret_cast<const float4*>(lg)[j4] : make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j4 = base + u * 64 + lane;
                if (j4 >= n4) continue;
                if (lo) reinterpret_cast<float4*>(lo)[j4] = v[u];
                const int j = j4 * 4;                          // ascending index: first maximum wins inside a lane
                if (v[u].x > best) { best = v[u].x; bi = j; }
                if (v[u].y > best) { best = v[u].y; bi = j + 1; }
                if (v[u].z > best) { best = v[u].z; bi = j + 2; }
                if (v[u].w > best) { best = v[u].w; bi = j + 3; }
            }
        }
    } else {
        for (int j = lane; j < a.V; j += 64) {
            const float v = lg[j];
            if (lo) lo[j] = v;
            if (v > best) { best = v; bi = j; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }   // ties -> lowest index (torch.argmax)
    }
    if (lane == 0) {
        a.cur_tok[row] = bi;
        if (a.tokens_out) a.tokens_out[(size_t)row * a.out_stride + t] = bi;
        unsigned add = 1u;
        if (a.eos >= 0 && bi == a.eos && !a.eos_seen[row]) { a.eos_seen[row] = 1; add += 1u << 16; }
        const unsigned old = atomicAdd(&a.st->arrive, add);
        if ((old & 0xffffu) == (unsigned)(a.rows - 1)) {          // last arriver: sole owner of the state now
            const int total = a.st->rows_with_eos + (int)((old + add) >> 16);
            a.st->rows_with_eos = total;
            a.done_flag[t] = (a.eos >= 0 && total >= a.rows) ? 1 : 0;
            a.st->arrive = 0u;
            a.st->t = t + 1;
        }
    }
}

// (re)start a decode: position 0, BOS everywhere; a BOS that equals eos already satisfies the check
__global__ void reset_state_kernel(StepState* st, int64_t* cur_tok, int* eos_seen, int* done_flag, int rows,
                                   int n_flags, int bos, int eos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) { cur_tok[i] = bos; eos_seen[i] = (eos >= 0 && bos == eos) ? 1 : 0; }
    if (i < n_flags) done_flag[i] = 0;
    if (i == 0) { st->t = 0; st->rows_with_eos = (eos >= 0 && bos == eos) ? rows : 0; st->arrive = 0u; st->pad = 0; }
}

__global__ void set_position_kernel(StepState* st, int t) { st->t = t; st->arrive = 0u; }

template <typename T>
__global__ void cast_rows_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(out)[i] = v;
    else {
        union { bf16 h[4]; uint2 u; } c;
        c.h[0] = __float2bfloat16(v.x); c.h[1] = __float2bfloat16(v.y);
        c.h[2] = __float2bfloat16(v.z); c.h[3] = __float2bfloat16(v.w);
        reinterpret_cast<uint2*>(out)[i] = c.u;
    }
}

}  // namespace txo
