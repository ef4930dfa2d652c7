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
    // sampling (reference decoder.py:104-108): keep the topk largest logits, softmax(logits / temp), draw one
    int topk; float inv_temp; unsigned long long seed;
    int row0;                   // first row of this row range in the batch: a draw is keyed by (seed; row of the BATCH, t), whatever the ranges
    // per-row stop (txo_set_stop_mode: a build extension, the reference only breaks globally): a row that already contains eos is frozen --
    // nothing more is written for it -- and the live rows of a range may have been compacted to its front (Engine::compact_lane):
    // row_map[row] = row of the BATCH that sits in slot `row` of the range (null: the identity).  Outputs and sampler keys follow it.
    int stop_rows; const int* row_map;
};
// row of the range (relative to row0) whose outputs slot `row` produces
__device__ inline int out_row(const StepArgs& a, int row) { return a.row_map ? a.row_map[row] - a.row0 : row; }

// append the chosen token, update the GLOBAL eos bookkeeping, advance the device-side position (lane 0 of a row)
// a token id a selection can produce only from non-finite logits (arg-max over NaNs finds nothing: index 0x7fffffff) must never reach
// the next position's embedding lookup: ids are forced into the vocabulary (txo_encode: non-finite input gives unspecified, in-range tokens)
__device__ inline int in_vocab(int tok, int V) { return (unsigned)tok < (unsigned)V ? tok : 0; }
__device__ inline void commit_token(const StepArgs& a, int row, int t, int tok) {
    tok = in_vocab(tok, a.V);
    a.cur_tok[row] = tok;
    const bool frozen = a.stop_rows && a.eos >= 0 && a.eos_seen[row];          // per-row stop: the host pads behind the row's first eos
    if (a.tokens_out && !frozen) a.tokens_out[(size_t)out_row(a, row) * a.out_stride + t] = tok;
    unsigned add = 1u;
    if (a.eos >= 0 && tok == a.eos && !a.eos_seen[row]) { a.eos_seen[row] = 1; add += 1u << 16; }
    const unsigned old = atomicAdd(&a.st->arrive, add);
    if ((old & 0xffffu) == (unsigned)(a.rows - 1)) {          // last arriver: sole owner of the state now
        const int total = a.st->rows_with_eos + (int)((old + add) >> 16);
        a.st->rows_with_eos = total;
        a.done_flag[t] = (a.eos >= 0 && total >= a.rows) ? 1 : 0;
        a.st->arrive = 0u;
        a.st->t = t + 1;
    }
}

__global__ __launch_bounds__(64) void argmax_step_kernel(StepArgs a) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int t = a.st->t;
    const float* lg = a.logits + (size_t)row * a.V;
    float best = -3.4e38f; int bi = 0x7fffffff;
    float* lo = a.logits_out ? a.logits_out + ((size_t)out_row(a, row) * a.out_stride + t) * a.V : nullptr;
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
    wave_argmax(best, bi);                                                 // ties -> lowest index (torch.argmax)
    if (lane == 0) commit_token(a, row, t, bi);
}

// Philox4x32-10 (counter-based, no state to carry between steps): counter = (row, t, 0, 0), key = seed.
__device__ inline void philox4x32(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ inline unsigned fkey(float f) {          // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The reference sampler for one step (decoder.py:104-108, utils.py:85-91): top-k filter (k = int((1-0.9)*V)),
// softmax(logits / temp), one multinomial draw.  One wave per row, the row staged in LDS (row_lds: V floats owned by this wave).
//   k-th largest: bitwise bisection on order-preserving uint keys (32 counting passes);
//   draw: u ~ Philox keyed by (seed; row, t) -> inverse CDF over the kept entries in index order (lane-contiguous chunks + wave scan).
// Statistically equivalent to torch.multinomial (a different RNG stream), reproducible for a given seed.
// Shared by sample_step_kernel (launch path) and the persistent decode kernel (persist.h; LOAD reads the logits another
// workgroup of the same launch wrote): same operations in the same order -> the same draw for the same (seed, row, t).
template <class Load>
__device__ inline int sample_row_lds(Load&& load, float* row_lds, float* lo, int V, int lane, int topk, float inv_temp,
                                 unsigned long long seed, unsigned row, unsigned t) {
    float mx = -3.4e38f;
    for (int j = lane; j < V; j += 64) { const float v = load(j); row_lds[j] = v; if (lo) lo[j] = v; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    __builtin_amdgcn_s_waitcnt(0xC07F);              // own LDS writes done (single wave)
    // k-th largest key
    const int k = min(max(topk, 1), V);
    unsigned prefix = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        int cnt = 0;
        for (int j = lane; j < V; j += 64) cnt += fkey(row_lds[j]) >= cand ? 1 : 0;
        cnt = wave_sum(cnt);
        if (cnt >= k) prefix = cand;
    }
    // kept set: keys > prefix, plus the first (k - n_greater) entries equal to it in index order (ties)
    const int per = (V + 63) / 64, j0 = lane * per, j1 = min(V, j0 + per);
    int ngt = 0, neq = 0;
    for (int j = j0; j < j1; ++j) { const unsigned key = fkey(row_lds[j]); ngt += key > prefix; neq += key == prefix; }
    int ngt_all = ngt, eq_before = neq;
    ngt_all = wave_sum(ngt_all);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(eq_before, o, 64); if (lane >= o) eq_before += v; }
    eq_before -= neq;                                 // exclusive scan of the tie counts
    int eq_left = (k - ngt_all) - eq_before;          // ties this lane may still keep
    float psum = 0.f;
    for (int j = j0; j < j1; ++j) {
        const float v = row_lds[j];
        const unsigned key = fkey(v);
        bool keep = key > prefix;
        if (key == prefix && eq_left > 0) { keep = true; --eq_left; }
        const float p = keep ? expf((v - mx) * inv_temp) : 0.f;
        row_lds[j] = p;                               // this lane owns [j0, j1)
        psum += p;
    }
    float incl = psum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const float total = __shfl(incl, 63, 64);
    unsigned c[4] = {row, t, 0u, 0u};
    philox4x32(c, (unsigned)seed, (unsigned)(seed >> 32));
    const float u = ((c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);    // (0, 1)
    const float target = u * total;
    // the lane whose inclusive prefix first reaches the target holds the sample
    const unsigned long long hit = __ballot(incl >= target && psum > 0.f);
    const int owner = hit ? (int)__builtin_ctzll(hit) : 63;
    int pick = -1;
    if (lane == owner) {
        float run = incl - psum;
        for (int j = j0; j < j1; ++j) { const float p = row_lds[j]; if (p > 0.f) { pick = j; run += p; if (run >= target) break; } }
    }
    pick = __shfl(pick, owner, 64);
    if (pick < 0) {                                   // numerical corner (target beyond the last kept entry): take the arg max
        float best = -3.4e38f; int bi = 0x7fffffff;
        for (int j = lane; j < V; j += 64) { const float v = load(j); if (v > best) { best = v; bi = j; } }
        wave_argmax(best, bi);
        pick = bi;
    }
    return pick;
}

// The same sampler with the row in REGISTERS (vocabularies up to 64 * SR_PER entries): the LDS form above spends ~30 us per row in
// dependent LDS round trips (32 bisection passes x 16 strided reads per lane); here every lane keeps its SR_PER consecutive logits,
// a bisection pass counts with one ballot + scalar popcount per register slot (the count is wave-uniform, no cross-lane sum), and the
// bisection stops at the first candidate that separates exactly k entries (any threshold with k entries at or above it gives the same
// kept set).  Same kept set, same probabilities summed in the same order, same counter RNG key -> the SAME draw as the LDS form.
#ifndef TXO_SAMPLER_LDS
#define TXO_SAMPLER_LDS 0      // 1: build the LDS form only (tests of its equality with the register form: probes/ab_libs.sh)
#endif
constexpr int SR_PER = 16;
// load4(j): four consecutive logits from index j (a multiple of 4) as one 16-byte request (their bits as u32x4); used when V % 4 == 0.
// r05: the row arrives in 16-byte requests, the probabilities are computed ONCE, without a branch per slot, and kept in the key
// registers for the owner's walk (the walk used to recompute them behind 16 more exec-mask branches): sample_step_kernel 13.0 -> 9.7 us
// per launch (rocprofv3, batch 64), the sampling stage of the persistent launch 10.7 -> 7.4 us by its stamps -- and the generate of the
// persistent launch unchanged (35.1 ms at batch 64 before and after: profiles/r05_sampler.txt).  Kept set, probabilities, summation
// order and RNG key are unchanged: the same draws (token hashes equal).
template <class Load, class Load4>
__device__ __forceinline__ int sample_row_regs(Load&& load, Load4&& load4, float* lo, int V, int lane, int topk, float inv_temp,
                                               unsigned long long seed, unsigned row, unsigned t) {
    // (inlined on purpose: as a called function it put the persistent decode kernel under the calling convention -- values live
    // across the call site pinned to callee-saved registers -- and cost the GREEDY decode 1.5 %; ONE 16-register array keeps it
    // inside that kernel's budget)
    // (the lane index made opaque: inside the persistent kernel's position loop everything below that depends only on the lane and V -- 16 slot
    // masks in SGPR pairs -- is otherwise hoisted out of the loop and held across all of its stages)
    asm volatile("" : "+v"(lane));
    const int per = (V + 63) / 64, j0 = lane * per;           // the LDS form's lane-contiguous chunks [j0, j1)
    // ONE register array: the order-preserving keys (a logit is recovered from its key by unfkey), later the probabilities
    auto unfkey = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
    unsigned key[SR_PER];
    float mx = -3.4e38f;
    // (no branch on a lane's own condition anywhere below: requests are clamped, results masked -- hipcc otherwise wraps every slot in an
    // exec-mask branch and copies the whole key array at each of them)
    const int nin = max(0, min(per, V - j0));                 // this lane's real slots
    if ((V & 3) == 0 && (per & 3) == 0) {                     // (wave-uniform) whole 16-byte pieces: j0 and V are multiples of 4
#pragma unroll
        for (int c = 0; c < SR_PER / 4; ++c) {
            const bool in = 4 * c < nin;
            const u32x4 w = load4(in ? j0 + 4 * c : 0);
            const float vx = __uint_as_float(w.x), vy = __uint_as_float(w.y), vz = __uint_as_float(w.z), vw = __uint_as_float(w.w);
            const unsigned m = in ? 0xffffffffu : 0u;         // 0 lies below every real key
            key[4 * c + 0] = fkey(vx) & m; key[4 * c + 1] = fkey(vy) & m; key[4 * c + 2] = fkey(vz) & m; key[4 * c + 3] = fkey(vw) & m;
            mx = fmaxf(mx, in ? fmaxf(fmaxf(vx, vy), fmaxf(vz, vw)) : -3.4e38f);
        }
    } else {
#pragma unroll
        for (int u = 0; u < SR_PER; ++u) {
            const bool in = u < nin;
            const float v = load(in ? j0 + u : 0);
            key[u] = fkey(v) & (in ? 0xffffffffu : 0u);
            mx = fmaxf(mx, in ? v : -3.4e38f);
        }
    }
    if (lo) {                                                 // (wave-uniform) the caller wants the logits of this position
        float* dst = lo + j0;
#pragma unroll
        for (int u = 0; u < SR_PER; ++u) { if (u < nin) dst[u] = unfkey(key[u]); }
    }
    mx = wave_max(mx);
    const int k = min(max(topk, 1), V);
    unsigned prefix = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < SR_PER; ++u) cnt += (int)__builtin_popcountll(__ballot(key[u] >= cand));
        if (cnt >= k) prefix = cand;
        if (cnt == k) break;                                  // exactly k entries at or above cand: the kept set is fixed
    }
    // slots beyond the row hold key 0: never above a prefix, and equal to it only when prefix == 0
    int ngt = 0, neq = 0;
#pragma unroll
    for (int u = 0; u < SR_PER; ++u) { ngt += key[u] > prefix; neq += (key[u] == prefix) && u < nin; }
    int ngt_all = wave_sum(ngt), eq_before = neq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(eq_before, o, 64); if (lane >= o) eq_before += v; }
    eq_before -= neq;
    const int eq_mine = (k - ngt_all) - eq_before;            // ties this lane may keep, in index order
    // probabilities (0 when not kept), computed for every slot and selected: no branch around the exponential
    float psum = 0.f;
    {
        int ties = 0;
#pragma unroll
        for (int u = 0; u < SR_PER; ++u) {
            const bool in = u < nin;
            const unsigned ku = key[u];
            const bool eq = in && ku == prefix;
            const bool keep = in && (ku > prefix || (eq && ties < eq_mine));
            ties += eq ? 1 : 0;
            const float e = expf(((in ? unfkey(ku) : mx) - mx) * inv_temp);
            const float pu = keep ? e : 0.f;
            key[u] = __float_as_uint(pu);
            psum += pu;                                       // (+ 0 for a slot that is not kept or not there: the same bits as skipping it)
        }
    }
    float incl = psum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const float total = __shfl(incl, 63, 64);
    unsigned c[4] = {row, t, 0u, 0u};
    philox4x32(c, (unsigned)seed, (unsigned)(seed >> 32));
    const float uu = ((c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);    // (0, 1)
    const float target = uu * total;
    const unsigned long long hit = __ballot(incl >= target && psum > 0.f);
    const int owner = hit ? (int)__builtin_ctzll(hit) : 63;
    // every lane walks its own chunk (no branch); the owner's result is the draw
    int pick = -1;
    {
        float run = incl - psum;
        bool done = false;
#pragma unroll
        for (int u = 0; u < SR_PER; ++u) {
            const float pu = __uint_as_float(key[u]);
            const bool take = !done && pu > 0.f;
            pick = take ? j0 + u : pick;
            run = take ? run + pu : run;
            done = done || (take && run >= target);
        }
    }
    pick = __shfl(pick, owner, 64);
    if (pick < 0) {                                   // numerical corner (target beyond the last kept entry): take the arg max (the row is read again)
        float best = -3.4e38f; int bi = 0x7fffffff;
        for (int u = 0; u < nin; ++u) { const float v = load(j0 + u); if (v > best) { best = v; bi = j0 + u; } }
        wave_argmax(best, bi);
        pick = bi;
    }
    return pick;
}

// REGS: the row in registers (host: vocabularies up to 64 * SR_PER entries) or staged in LDS.  Two kernels, not one with a branch: with both
// forms in one body hipcc keeps a 36-byte stack object (private segment enabled for every launch) that neither form has alone.
__host__ __device__ inline bool sample_in_regs(int V) { return V <= 64 * SR_PER && !TXO_SAMPLER_LDS; }
template <bool REGS>
__global__ __launch_bounds__(64) void sample_step_kernel(StepArgs a) {
    extern __shared__ float row_lds[];
    const int row = blockIdx.x, lane = threadIdx.x, V = a.V;
    const int t = a.st->t;
    const float* lg = a.logits + (size_t)row * V;
    const int orow = out_row(a, row);
    float* lo = a.logits_out ? a.logits_out + ((size_t)orow * a.out_stride + t) * V : nullptr;
    int pick;
    if constexpr (REGS) pick = sample_row_regs([&](int j) { return lg[j]; }, [&](int j) { return ld16(lg + j); }, lo, V, lane, a.topk, a.inv_temp, a.seed,
                                               (unsigned)(a.row0 + orow), (unsigned)t);
    else pick = sample_row_lds([&](int j) { return lg[j]; }, row_lds, lo, V, lane, a.topk, a.inv_temp, a.seed, (unsigned)(a.row0 + orow), (unsigned)t);
    if (lane == 0) commit_token(a, row, t, pick);
}
inline void launch_sample_step(hipStream_t s, int rows, const StepArgs& sa) {
    if (sample_in_regs(sa.V)) hipLaunchKernelGGL(sample_step_kernel<true>, dim3(rows), dim3(64), 0, s, sa);
    else hipLaunchKernelGGL(sample_step_kernel<false>, dim3(rows), dim3(64), (size_t)sa.V * sizeof(float), s, sa);
}

// ---- per-row stop (a build extension; the reference breaks only globally, decoder.py:115-116) -------------------
// Rows are independent (the only cross-row operation of the path is that global test), so a row that has produced eos needs no further
// work: its later tokens are `pad`.  Two pieces: (1) pad_after_eos_kernel rewrites everything behind a row's first eos once the decode is
// over (every decode path); (2) on the launch path the live rows of a row range are compacted to the front of the range every few positions
// and the range's launches shrink -- compact_scan_kernel builds the new small state and the list of row moves, move_rows_kernel moves the
// rows of the K/V history and of the cross-attention operand (same bits per row: nothing a row computes depends on its slot).
__global__ void pad_after_eos_kernel(int64_t* tokens, int stride, int steps, int rows, int eos, int bos, int pad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int64_t* row = tokens + (size_t)r * stride;
    bool done = bos == eos;                                   // the BOS column counts (decoder.py:115 looks at the whole output)
    for (int t = 0; t < steps; ++t) {
        if (done) row[t] = pad;
        else if (row[t] == eos) done = true;
    }
}

// txo_decode_step's tok_in: ids outside [0, vocab) are forced into the table (the reference's nn.Embedding would raise; the Python facade does)
__global__ void copy_tokens_kernel(int64_t* dst, const int64_t* src, int n, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const long long v = src[i]; dst[i] = v < 0 ? 0 : (v >= V ? V - 1 : v); }
}
__global__ void iota_kernel(int* p, int n, int first) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = first + i; }

struct CompactArgs {
    int rows, new_rows, row0;             // rows of the range now / after (new_rows >= live rows: the host's bound is a few positions old)
    int64_t* cur_tok; int* eos_seen; int* row_map;            // the range's slices, rewritten in place
    int64_t* cur_tok2; int* row_map2;                         // scratch [rows]
    int* moves;                           // out: (src, dst) pairs of the live rows that change slot, ascending; info = {live rows, moves}
    int* info; StepState* st; int fill_tok;
};
// ONE workgroup.  Live rows keep their order, so dst <= src, and a live row moves iff a finished row precedes it: with F = the first
// finished row (every row before it is live and stays), the live row that lands in slot dst > = F is entry dst - F of the move list.
__global__ __launch_bounds__(1024) void compact_scan_kernel(CompactArgs a) {
    __shared__ int wsum[16], s_base, s_first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_base = 0; s_first = a.rows; }
    __syncthreads();
    for (int c0 = 0; c0 < a.rows; c0 += 1024) {
        const int r = c0 + tid;
        const int live = (r < a.rows && !a.eos_seen[r]) ? 1 : 0;
        int inc = live;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (lane == 63) wsum[wave] = inc;
        if (r < a.rows && !live) atomicMin(&s_first, r);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int dst = before + inc - live, F = s_first;     // F is final for every row up to this chunk's end
        if (live) {
            a.cur_tok2[dst] = a.cur_tok[r]; a.row_map2[dst] = a.row_map[r];
            if (dst != r) { a.moves[2 * (dst - F)] = r; a.moves[2 * (dst - F) + 1] = dst; }
        }
        __syncthreads();
        if (tid == 0) { int s = s_base; for (int w = 0; w < 16; ++w) s += wsum[w]; s_base = s; }
        __syncthreads();
    }
    const int L = s_base, F = min(s_first, L);
    // slots [L, new_rows): filler rows that count as finished (the host's row count is an upper bound a few positions old)
    for (int i = tid; i < a.new_rows; i += 1024) {
        const bool lv = i < L;
        a.cur_tok[i] = lv ? a.cur_tok2[i] : (int64_t)a.fill_tok;
        a.row_map[i] = lv ? a.row_map2[i] : a.row0;
        a.eos_seen[i] = lv ? 0 : 1;
    }
    if (tid == 0) { a.info[0] = L; a.info[1] = L - F; a.st->rows_with_eos = a.new_rows - L; a.st->arrive = 0u; }
}

// Moves rows of a [outer][inner_n][rows][...] family of planes: plane p = (o, i) starts at base + o * outer_stride + i * inner_stride (bytes),
// row r of it at + r * row_stride, and the first `len16` 16-byte pieces of a row are copied from moves[2k] to moves[2k+1], k ascending.
// A thread owns ONE piece offset of ONE plane and walks the moves in order, eight loads ahead of their stores: every store of a group goes
// to a row <= the group's last source, and every live row up to there has been read (by this same thread, in program order) -- in place, no
// barrier.  blockIdx.y = plane.
struct MoveArgs { unsigned char* base; size_t outer_stride, inner_stride, row_stride; int inner_n; unsigned len16; const int* moves; const int* info; };
__global__ __launch_bounds__(256) void move_rows_kernel(MoveArgs a) {
    const unsigned piece = blockIdx.x * 256u + threadIdx.x;
    if (piece >= a.len16) return;
    const int p = blockIdx.y, o = p / a.inner_n, i = p - o * a.inner_n;
    unsigned char* pl = a.base + (size_t)o * a.outer_stride + (size_t)i * a.inner_stride + (size_t)piece * 16;
    const int n = a.info[1];
    for (int k0 = 0; k0 < n; k0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = min(k0 + u, n - 1); v[u] = ld16(pl + (size_t)a.moves[2 * k] * a.row_stride); }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < n) st16(pl + (size_t)a.moves[2 * (k0 + u) + 1] * a.row_stride, v[u]);
    }
}

// ---- beam search (a build extension: the reference has no beam search) -----------------------------------
// Length-unnormalised sum of log_softmax(logits); k beams per image, only beam 0 live at step 0; a beam that has
// emitted eos is finished and continues with eos at no cost; candidates ranked by score, ties -> lower flat index
// (beam * V + token); the loop stops when every beam of every image is finished.
struct BeamArgs {
    const float* logits; int V, k, images;
    int64_t* cur_tok;                     // [images*k] token fed to the next step
    float* score; int* fin;               // [images*k] running log-prob, finished flag (updated in place)
    const short* path_cur; short* path_nxt; int path_stride;   // [rows][tmax] slot of every history position
    short* parent_hist; int* tok_hist; int hist_stride;        // [tmax][rows] back-pointers for the final backtrack
    StepState* st; int* done_flag; int eos;
    int row0;                             // this row range's first row in the batch: every pointer above is the RANGE's (slots in `path` are range-local,
                                          // like the range's K/V base), only the back-pointers name rows of the whole batch (beam_backtrack_kernel)
};

__global__ __launch_bounds__(256) void beam_select_kernel(BeamArgs a) {
    constexpr int KMAX = 8;
    __shared__ float red[8], s_lse[KMAX], s_score[KMAX], selv[KMAX];
    __shared__ int redi[4], s_fin[KMAX], seli[KMAX];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k, V = a.V, t = a.st->t;
    const float* lg = a.logits + (size_t)img * k * V;
    constexpr int VPT = 4;                                    // register path: vocabularies up to 256 * VPT entries
    if (V <= 256 * VPT) {
        // The image's k rows live in registers (thread tid holds entries tid + 256 i of every row): ONE read of the logits, the k
        // log-sum-exps reduced together (two barriers instead of three per beam), k selection rounds over registers (one barrier
        // each).  Same per-thread summation order, same wave / block reductions and the same tie rule as the general path below
        // -> the same bits (it measured 50 us per position at 128 images x 5 beams x 1000 entries against 640 rows' decode step).
        __shared__ float rmx[KMAX][4], rse[KMAX][4], rbv[2][4];
        __shared__ int rbi[2][4];
        float x[KMAX][VPT], sc[KMAX]; int fn[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            sc[j] = 0.f; fn[j] = 0;
            if (j < k) { sc[j] = a.score[img * k + j]; fn[j] = a.fin[img * k + j]; }
#pragma unroll
            for (int i = 0; i < VPT; ++i) { const int v = tid + 256 * i; x[j][i] = (j < k && v < V) ? lg[(size_t)j * V + v] : 0.f; }
        }
        float mx[KMAX], lse[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j < k) {
            float m = -3.4e38f;
#pragma unroll
            for (int i = 0; i < VPT; ++i) if (tid + 256 * i < V) m = fmaxf(m, x[j][i]);
            m = wave_max(m);
            if (lane == 0) rmx[j][wave] = m;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j < k) {
            mx[j] = fmaxf(fmaxf(rmx[j][0], rmx[j][1]), fmaxf(rmx[j][2], rmx[j][3]));
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < VPT; ++i) if (tid + 256 * i < V) se += expf(x[j][i] - mx[j]);
            se = wave_sum(se);
            if (lane == 0) rse[j][wave] = se;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j < k) lse[j] = mx[j] + logf((rse[j][0] + rse[j][1]) + (rse[j][2] + rse[j][3]));
        // candidates in place of the logits
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j < k) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = tid + 256 * i;
                if (fn[j]) x[j][i] = (v == a.eos) ? sc[j] : -INFINITY;
                else x[j][i] = sc[j] + (x[j][i] - lse[j]);
            }
        }
        unsigned taken = 0u;                                  // bit j * VPT + i: this thread's candidate was selected in an earlier round
        float sv[KMAX]; int si[KMAX];
#pragma unroll
        for (int r = 0; r < KMAX; ++r) if (r < k) {
            float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) if (j < k) {
#pragma unroll
                for (int i = 0; i < VPT; ++i) {
                    const int v = tid + 256 * i, f = j * V + v;
                    if (v >= V || ((taken >> (j * VPT + i)) & 1u)) continue;
                    const float c = x[j][i];
                    if (c > best || (c == best && f < bi)) { best = c; bi = f; }
                }
            }
            wave_argmax(best, bi);
            if (lane == 0) { rbv[r & 1][wave] = best; rbi[r & 1][wave] = bi; }
            __syncthreads();
            float b = rbv[r & 1][0]; int ix = rbi[r & 1][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const float bw = rbv[r & 1][w]; const int iw = rbi[r & 1][w]; if (bw > b || (bw == b && iw < ix)) { b = bw; ix = iw; } }
            sv[r] = b; si[r] = ix;
            const int j = ix / V, v = ix - j * V;
            if ((v & 255) == tid) taken |= 1u << (j * VPT + (v >> 8));
        }
        int allfin = 1;
#pragma unroll
        for (int r = 0; r < KMAX; ++r) if (r < k) {
            const int sel = (unsigned)si[r] < (unsigned)(k * V) ? si[r] : 0;         // (non-finite logits select nothing: stay inside the image's candidates)
            const int j = sel / V, v = sel - j * V;
            int fj = 0;
#pragma unroll
            for (int q = 0; q < KMAX; ++q) if (q == j) fj = fn[q];
            const int nf = fj | ((a.eos >= 0 && v == a.eos) ? 1 : 0);
            allfin &= nf;
            if (tid == r) {
                const int row = img * k + r;
                a.score[row] = sv[r]; a.fin[row] = nf; a.cur_tok[row] = v;
                a.tok_hist[(size_t)t * a.hist_stride + row] = v;
                a.parent_hist[(size_t)t * a.hist_stride + row] = (short)(a.row0 + img * k + j);
            }
            const short* src = a.path_cur + (size_t)(img * k + j) * a.path_stride;
            short* dst = a.path_nxt + (size_t)(img * k + r) * a.path_stride;
            for (int p2 = tid; p2 < t; p2 += 256) dst[p2] = src[p2];
            if (tid == 0) dst[t] = (short)(img * k + j);
        }
        if (tid == 0) {
            const unsigned add = 1u + ((a.eos >= 0 && allfin) ? (1u << 16) : 0u);
            const unsigned old = atomicAdd(&a.st->arrive, add);
            if ((old & 0xffffu) == (unsigned)(a.images - 1)) {
                a.done_flag[t] = (int)((old + add) >> 16) >= a.images ? 1 : 0;
                a.st->arrive = 0u;
                a.st->t = t + 1;
            }
        }
        return;
    }
    if (tid < k) { s_score[tid] = a.score[img * k + tid]; s_fin[tid] = a.fin[img * k + tid]; }
    // log-sum-exp of every beam's row
    for (int j = 0; j < k; ++j) {
        float mx = -3.4e38f;
        for (int v = tid; v < V; v += 256) mx = fmaxf(mx, lg[(size_t)j * V + v]);
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float se = 0.f;
        for (int v = tid; v < V; v += 256) se += expf(lg[(size_t)j * V + v] - mx);
        se = wave_sum(se);
        if (lane == 0) red[4 + wave] = se;
        __syncthreads();
        if (tid == 0) s_lse[j] = mx + logf((red[4] + red[5]) + (red[6] + red[7]));
        __syncthreads();
    }
    // k rounds of block-wide arg max over the k*V candidates, excluding what was already taken
    const int total = k * V;
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int f = tid; f < total; f += 256) {
            bool taken = false;
            for (int q = 0; q < r; ++q) taken |= seli[q] == f;
            if (taken) continue;
            const int j = f / V, v = f - j * V;
            float c;
            if (s_fin[j]) c = (v == a.eos) ? s_score[j] : -INFINITY;
            else c = s_score[j] + (lg[f] - s_lse[j]);
            if (c > best || (c == best && f < bi)) { best = c; bi = f; }
        }
        wave_argmax(best, bi);
        if (lane == 0) { red[wave] = best; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float b = red[0]; int i = redi[0];
            for (int w = 1; w < 4; ++w) if (red[w] > b || (red[w] == b && redi[w] < i)) { b = red[w]; i = redi[w]; }
            selv[r] = b; seli[r] = i;
        }
        __syncthreads();
    }
    // new beams: slot r of this image continues beam seli[r] / V with token seli[r] % V
    int allfin = 1;
    for (int r = 0; r < k; ++r) {
        const int sel = (unsigned)seli[r] < (unsigned)(k * V) ? seli[r] : 0;         // (non-finite logits: see above)
        const int j = sel / V, v = sel - j * V;
        const int nf = s_fin[j] | ((a.eos >= 0 && v == a.eos) ? 1 : 0);
        allfin &= nf;
        if (tid == r) {
            const int row = img * k + r;
            a.score[row] = selv[r]; a.fin[row] = nf; a.cur_tok[row] = v;
            a.tok_hist[(size_t)t * a.hist_stride + row] = v;
            a.parent_hist[(size_t)t * a.hist_stride + row] = (short)(a.row0 + img * k + j);
        }
        // history slots of the new beam: the parent's, plus the parent's own slot for position t
        const short* src = a.path_cur + (size_t)(img * k + j) * a.path_stride;
        short* dst = a.path_nxt + (size_t)(img * k + r) * a.path_stride;
        for (int p = tid; p < t; p += 256) dst[p] = src[p];
        if (tid == 0) dst[t] = (short)(img * k + j);
    }
    if (tid == 0) {
        const unsigned add = 1u + ((a.eos >= 0 && allfin) ? (1u << 16) : 0u);
        const unsigned old = atomicAdd(&a.st->arrive, add);
        if ((old & 0xffffu) == (unsigned)(a.images - 1)) {
            a.done_flag[t] = (int)((old + add) >> 16) >= a.images ? 1 : 0;
            a.st->arrive = 0u;
            a.st->t = t + 1;
        }
    }
}

__global__ void beam_reset_kernel(StepState* st, int64_t* cur_tok, float* score, int* fin, int* done_flag, int rows, int k,
                                  int n_flags, int bos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) { cur_tok[i] = bos; score[i] = (i % k == 0) ? 0.f : -INFINITY; fin[i] = 0; }
    if (i < n_flags) done_flag[i] = 0;
    if (i == 0) { st->t = 0; st->rows_with_eos = 0; st->arrive = 0u; st->pad = 0; }
}

// follow the back-pointers from the last step: tokens_all[row][0..n)
__global__ void beam_backtrack_kernel(const short* parent_hist, const int* tok_hist, int hist_stride, int n, int rows,
                                      int64_t* tokens_all, int out_stride) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    int cur = row;
    for (int t = n - 1; t >= 0; --t) {
        tokens_all[(size_t)row * out_stride + t] = tok_hist[(size_t)t * hist_stride + cur];
        cur = parent_hist[(size_t)t * hist_stride + cur];
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

// holds one wave per workgroup for `ticks` x 10 ns (Engine::tune_lane_streams: do two streams' launches overlap?)
__global__ void hold_kernel(int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
}

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

// txo_decode_set_key_mask: the caller's (rows, cols) padding mask -> the engine's [rows][tmax] key mask (positions >= cols: attended)
__global__ void set_key_mask_kernel(const unsigned char* __restrict__ mask, unsigned char* __restrict__ kmask, int rows, int cols, int tmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * tmax) return;
    const int r = i / tmax, j = i - r * tmax;
    kmask[i] = j < cols ? (mask[(size_t)r * cols + j] != 0 ? 1 : 0) : 1;
}

}  // namespace txo
